"""Levelized gate netlists over a ciphertext pool (BASELINE config "levelized 32-bit adder /
comparator").

The reference has no netlist abstraction: its circuits are sequential loops of bootsXXX calls
(src/test/test-addition-boot.cpp:30-115).  Here a circuit is recorded as a gate list over numbered wires (= rows of
a ciphertext pool), sorted into dependency levels; every level is one batched call
(Engine.gate_level_dev -> tfhe_hip_gate_level_dev), and W independent instances of the circuit widen each level W
times.  The circuits below emit exactly the gate sequences of the reference's full_adder / full_adder_MUX /
comparison_MUX, so outputs are bit-identical to running those loops on the CPU.
"""
import numpy as np

from .engine import Gate

_NIN = {Gate.MUX: 3, Gate.NOT: 1, Gate.COPY: 1, Gate.CONST0: 0, Gate.CONST1: 0}


def n_inputs(gate):
    return _NIN.get(Gate(gate), 2)


class Netlist:
    def __init__(self):
        self.n_wires = 0
        self.gates = []          # (gate, out, a, b, c)
        self.inputs = {}         # name -> list of wires
        self.outputs = {}

    def new_wire(self):
        self.n_wires += 1
        return self.n_wires - 1

    def input(self, name, nbits):
        self.inputs[name] = [self.new_wire() for _ in range(nbits)]
        return self.inputs[name]

    def output(self, name, wires):
        self.outputs[name] = list(wires)

    def gate(self, gate, a=-1, b=-1, c=-1):
        out = self.new_wire()
        self.gates.append((int(gate), out, a, b, c))
        return out

    def levelize(self):
        """ASAP levels: a gate sits one level after its latest producer; wires are written once (SSA), so the
        'no row read and written in one level' rule of tfhe_hip_gate_level_dev holds by construction"""
        level_of = {}
        levels = []
        for g in self.gates:
            gate, out, a, b, c = g
            ins = (a, b, c)[:n_inputs(gate)]
            lv = 1 + max([level_of.get(w, 0) for w in ins], default=0)
            level_of[out] = lv
            while len(levels) < lv:
                levels.append([])
            levels[lv - 1].append(g)
        return levels

    def level_arrays(self, instances=1):
        """levels as int32[count*instances][5] arrays; instance k uses rows k*n_wires + wire"""
        out = []
        for lv in self.levels_cached():
            base = np.array(lv, np.int64)
            blocks = []
            for k in range(instances):
                blk = base.copy()
                sel = blk[:, 1:] >= 0
                blk[:, 1:][sel] += k * self.n_wires
                blocks.append(blk)
            out.append(np.concatenate(blocks).astype(np.int32))
        return out

    def levels_cached(self):
        if getattr(self, "_lv_n", None) != len(self.gates):
            self._lv = self.levelize()
            self._lv_n = len(self.gates)
        return self._lv

    # ---- evaluation ------------------------------------------------------------------------------------------
    def evaluate_plain(self, values):
        """plaintext semantics (bits), for tests; values: wire -> bit array"""
        from .truth import TRUTH
        v = dict(values)
        for gate, out, a, b, c in self.gates:
            z = np.zeros_like(next(iter(values.values())))
            v[out] = TRUTH[gate](v.get(a, z), v.get(b, z), v.get(c, z))
        return v

    def evaluate(self, level_fn, pool_rows_per_instance=None, instances=1):
        """runs every level through level_fn(gates int32[count][5]); returns number of bootstrapped gates"""
        total = 0
        for lv in self.level_arrays(instances):
            level_fn(lv)
            total += int((lv[:, 0] <= int(Gate.MUX)).sum())
        return total


# ---- the reference's test circuits -----------------------------------------------------------------------------
def full_adder(nl, x, y):
    """reference src/test/test-addition-boot.cpp:70-94 (XOR/AND carry)"""
    carry = nl.gate(Gate.CONST0)     # the reference encrypts a 0; a trivial 0 keeps the netlist input-free
    s = []
    for xi, yi in zip(x, y):
        t = nl.gate(Gate.XOR, xi, yi)
        s.append(nl.gate(Gate.XOR, t, carry))
        t1 = nl.gate(Gate.AND, xi, yi)
        t2 = nl.gate(Gate.AND, carry, t)
        carry = nl.gate(Gate.COPY, nl.gate(Gate.XOR, t1, t2))
    s.append(nl.gate(Gate.COPY, carry))
    return s


def full_adder_mux(nl, x, y):
    """reference src/test/test-addition-boot.cpp:30-67 (MUX carry)"""
    carry = nl.gate(Gate.CONST0)
    s = []
    for xi, yi in zip(x, y):
        t = nl.gate(Gate.XOR, xi, yi)
        s.append(nl.gate(Gate.XOR, t, carry))
        t1 = nl.gate(Gate.AND, xi, yi)
        carry = nl.gate(Gate.COPY, nl.gate(Gate.MUX, t, carry, t1))
    s.append(nl.gate(Gate.COPY, carry))
    return s


def comparison_mux(nl, x, y):
    """reference src/test/test-addition-boot.cpp:97-115: comp = (x <= y)"""
    carry = nl.gate(Gate.CONST1)
    for xi, yi in zip(x, y):
        t = nl.gate(Gate.XOR, xi, yi)
        carry = nl.gate(Gate.COPY, nl.gate(Gate.MUX, t, yi, carry))
    return [nl.gate(Gate.COPY, carry)]
