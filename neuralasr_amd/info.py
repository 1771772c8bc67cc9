"""Application identity used by the logger (reference: info.py)."""
app_name = 'NeuralASR'
version = '0.1'
