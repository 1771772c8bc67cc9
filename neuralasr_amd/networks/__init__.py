"""HIP implementations of NeuralASR's `networks/` plugin surface (reference: networks/)."""
