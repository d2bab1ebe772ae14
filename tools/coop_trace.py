"""per-phase clock totals of the FAST latency kernel (build with -DTFHE_HIP_COOP_TRACE=1, TFHE_HIP_LIB=<that build>):
one gate, workgroup 0 prints its waves' totals over the 630 CMux iterations"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tfhe_amd import Client, Engine, Gate
client = Client(); keys = client.keygen(42)
a = client.encrypt(keys, np.ones(1, np.uint8), seed=1)
eng = Engine(0); eng.set_arithmetic("fast"); eng.load_keys_coef(keys.bk_coef, keys.ks)
eng.gate_batch(Gate.NAND, a, a)
eng.sync(); eng.close()
