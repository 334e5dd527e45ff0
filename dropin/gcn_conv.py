"""Drop-in for the reference's gcn_conv.py (imported by model.py:7)."""
from cal_amd.gcn_conv import GCNConv  # noqa: F401
