"""Drop-in for the reference's train_causal.py (`from train_causal import train_causal_syn` main_syn.py:2,
`from train_causal import train_causal_real` main_real.py:2): same function names and argument order
(train_causal.py:11,63,162,202), running on the HIP path."""
from cal_amd.train_causal import (causal_loss, eval_acc_causal, train_causal_epoch,  # noqa: F401
                                  train_causal_real, train_causal_syn)
