"""The Baum-Welch driver of the package (cpecan-signal_amd/em.py) under the name the tests use."""
from cpecan_load import em as _em

_m = _em()
EXP_LEN = _m.EXP_LEN
shard = _m.shard
allreduce_expectations = _m.allreduce_expectations
m_step = _m.m_step
train = _m.train
gpu_e_step = _m.gpu_e_step
PersistentEStep = _m.PersistentEStep
