"""Drop-in for the reference's datasets.py (`from datasets import get_dataset`, main_real.py:1; datasets.py:11-48):
TU text files -> feature-expanded dataset.  ``sparse`` / ``pruning_percent`` are accepted for signature compatibility
(the reference's CAL configurations never change them)."""
from cal_amd import tu


def get_dataset(name, sparse=True, feat_str="deg+ak3+reall", root=None, pruning_percent=0):
    return tu.get_dataset(name, feat_str=feat_str, root=root)
