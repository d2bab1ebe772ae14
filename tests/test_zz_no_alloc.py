"""tfhe_hip_gate_batch_dev_noalloc (include/tfhe_hip.h): the entry point that allocates nothing.  Called through the
C-ABI and compared BIT FOR BIT with tfhe_hip_gate_batch_dev on the same inputs, on two calls in a row, the second with
a smaller input (the static array keeps the first call's contents), plus a third in another launch shape and the size
limit.  Collected last (file name) so that a failure here cannot hide the other GPU tests under `pytest -x`."""
import numpy as np
import pytest

from tfhe_amd import Gate, PARAMS as P, TfheHipError

pytestmark = pytest.mark.gpu


def _dev(engine, arr):
    p = engine.dev_alloc(arr.nbytes)
    engine.to_dev(p, arr)
    return p


def test_noalloc_entry_point_equals_gate_batch_dev(engine, client, keys):
    rng = np.random.default_rng(2025)
    first, second = 1000, 37                      # 1000: four waves per workgroup + unsplit keyswitch; 37: latency kernel + split
    bits = rng.integers(0, 2, (3, first)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], seed=900 + i) for i in range(3)]
    d = [_dev(engine, c) for c in ct]
    d_ref, d_out = engine.dev_alloc(ct[0].nbytes), engine.dev_alloc(ct[0].nbytes)
    try:
        for gate, count in ((Gate.XOR, first), (Gate.MUX, second), (Gate.NAND, 1), (Gate.MUX, 1024 - 24), (Gate.NOT, 5)):
            count = min(count, first)
            engine.gate_batch_dev(gate, d_ref, d[0], d[1], d[2], count=count)
            engine.sync()
            engine.to_dev(d_out, np.full((first, P.n + 1), 0x5a5a5a5a, np.int32))   # stale contents must not survive
            engine.gate_batch_dev_noalloc(gate, d_out, d[0], d[1], d[2], count=count)
            engine.sync()
            want = engine.from_dev(d_ref, (count, P.n + 1))
            got = engine.from_dev(d_out, (count, P.n + 1))
            assert np.array_equal(got, want), (gate, count)
            if gate == Gate.XOR:
                assert np.array_equal(client.decrypt(keys, got)[0], bits[0] ^ bits[1])
        with pytest.raises(TfheHipError, match="NOALLOC_MAX_GATES"):
            engine.gate_batch_dev_noalloc(Gate.NAND, d_out, d[0], d[1], None, count=1025)
        engine.gate_batch_dev_noalloc(Gate.NAND, d_out, d[0], d[1], None, count=0)     # empty batch: nothing to do
    finally:
        for p in d + [d_ref, d_out]:
            engine.dev_free(p)
