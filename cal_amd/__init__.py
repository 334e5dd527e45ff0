"""cal_amd -- MI355X-native (gfx950) implementation of the CAL causal-attention
message-passing hot path (yongduosui/CAL: gcn_conv.py, model.py,
train_causal.py).  Hand-written HIP kernels behind a C ABI
(include/cal_hip.h, cal_amd/csrc) with a thin Python host that keeps the
reference's nn.Module surface.  There is no CPU fallback: compute entry points
raise if libcalhip.so or a GPU is missing."""
__version__ = "0.1.0"
