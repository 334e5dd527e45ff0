"""Drop-in for the reference's model.py (imported by opts.py:2): re-exports the cal_amd models."""
from cal_amd.model import CausalGAT, CausalGCN, CausalGIN  # noqa: F401


def _out_of_scope(name):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError(
                "%s is a baseline net outside the accelerated hot path (SURVEY.md section 8); "
                "use the reference's own model.py for it" % name)
    _Missing.__name__ = name
    return _Missing


GCNNet, GINNet, GATNet = _out_of_scope("GCNNet"), _out_of_scope("GINNet"), _out_of_scope("GATNet")
