"""CPU tests: the C-ABI library loads and exports every symbol include/plonky2_mi355x.h declares, and the
product path fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "plonky2_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from plonky2_demo_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), "libplonky2_mi355x.so does not export %s" % n
        assert n in _lib.SIGNATURES, "binding table misses %s" % n
    assert set(_lib.SIGNATURES) <= set(names), "binding table has symbols the header does not declare"


def test_product_never_touches_the_oracle():
    """The shipped path must not import, link, include or dlopen anything under oracle/."""
    pkg = os.path.join(ROOT, "plonky2_demo_amd")
    bad = re.compile(r"liboracle|oracle_lib|#include\s*[\"<][^\">]*oracle/|^\s*(import|from)\s+oracle|gl_field\.hpp|gl_batch\.hpp", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".cpp", ".h", ".inc")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), "%s references the oracle" % f


def test_oracle_never_touches_the_product():
    """... and the checker shares no source with what it checks (its Poseidon tables are its own transcription of the
    reference's, oracle/make_tables.py): a bug in a product header cannot be common-mode."""
    orc_dir = os.path.join(ROOT, "oracle")
    for f in os.listdir(orc_dir):
        if f.endswith((".hpp", ".cpp", ".h")) or f == "Makefile":
            assert "plonky2_demo_amd" not in open(os.path.join(orc_dir, f)).read(), "%s reaches into the product" % f


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import plonky2_demo_amd as p
    with pytest.raises(p.Plonky2Mi355xError):
        p.Context(0)


# ---- host-side circuit builder of the product (no GPU) against the oracle's independent restatement ----------
@pytest.mark.parametrize("m", [1, 2, 3, 5, 8, 20])
def test_host_matmul_circuit_matches_oracle(orc, m):
    import numpy as np
    import plonky2_demo_amd as p
    from oracle_lib import rand_field
    hc = p.MatmulCircuit(m)
    oc = orc.circuit(m, threads=4)
    assert hc.degree_bits == oc.info["degree_bits"]
    assert hc.desc.num_constants == oc.info["num_constants"] and hc.desc.num_selectors == oc.info["num_selectors"]
    assert hc.desc.num_fri_rounds == oc.info["num_fri_rounds"] and hc.desc.num_public_inputs == oc.info["num_public_inputs"]
    assert list(hc.desc.gate_types[: hc.desc.num_gates]) == oc.gate_order()
    assert (hc.row_gates() == oc.row_gates()).all()
    # constants + the 80 sigma polynomials: two independent constructions (closed-form classes vs union-find forest)
    assert (hc.constants_sigmas() == oc.constants_sigmas()).all()
    a, b = rand_field(m, m * m) % (2**32 - 1), rand_field(m + 100, m * m) % (2**32 - 1)
    wires, pis = hc.witness(a, b, filler_seed=99)
    ow = oc.witness(a, b, filler_seed=99)
    assert (pis == ow.public_inputs()).all()
    assert (wires == ow.wires()).all()


def test_challenger_matches_the_duplex_sponge_model(orc):
    # gl_challenger_* (iop/challenger.rs:30-153) against a Python model on the oracle's permutation: overwrite-mode duplexing,
    # challenges popped from the END of the rate, outputs discarded by any new observation; no GPU needed
    import numpy as np
    import plonky2_demo_amd as p
    from oracle_lib import P

    class Model:
        def __init__(self):
            self.state, self.inp, self.out = [0] * 12, [], []

        def dup(self):
            for i, x in enumerate(self.inp):
                self.state[i] = x
            self.inp = []
            self.state = [int(x) for x in orc.poseidon(np.array(self.state, dtype=np.uint64))]
            self.out = self.state[:8]

        def observe(self, xs):
            for x in xs:
                self.out = []
                self.inp.append(int(x))
                if len(self.inp) == 8:
                    self.dup()

        def get(self, k):
            r = []
            for _ in range(k):
                if self.inp or not self.out:
                    self.dup()
                r.append(self.out.pop() % P)
            return r

    ch, mo = p.Challenger(), Model()
    rng = np.random.default_rng(1)
    for _ in range(60):
        xs = rng.integers(0, 2**64, int(rng.integers(0, 20)), dtype=np.uint64)      # non-canonical inputs too
        ch.observe_elements(xs)
        mo.observe(xs)
        k = int(rng.integers(0, 5))
        assert ch.get_n_challenges(k) == mo.get(k)
        st, buf = ch.state()
        assert [int(x) % P for x in st] == [x % P for x in mo.state] and [int(x) for x in buf] == mo.inp
