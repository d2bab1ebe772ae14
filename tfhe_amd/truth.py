"""Truth tables of the 15 gate ids of include/tfhe_hip.h (reference src/libtfhe/boot-gates.cpp)."""
import numpy as np

TRUTH = {
    0: lambda a, b, c: 1 - (a & b),      # NAND
    1: lambda a, b, c: a | b,            # OR
    2: lambda a, b, c: a & b,            # AND
    3: lambda a, b, c: a ^ b,            # XOR
    4: lambda a, b, c: 1 - (a ^ b),      # XNOR
    5: lambda a, b, c: 1 - (a | b),      # NOR
    6: lambda a, b, c: (1 - a) & b,      # ANDNY
    7: lambda a, b, c: a & (1 - b),      # ANDYN
    8: lambda a, b, c: (1 - a) | b,      # ORNY
    9: lambda a, b, c: a | (1 - b),      # ORYN
    10: lambda a, b, c: np.where(a == 1, b, c),  # MUX
    11: lambda a, b, c: 1 - a,           # NOT
    12: lambda a, b, c: a,               # COPY
    13: lambda a, b, c: np.ones_like(a),   # CONSTANT 1
    14: lambda a, b, c: np.zeros_like(a),  # CONSTANT 0
}
