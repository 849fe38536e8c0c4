"""ctypes binding of libplonky2_mi355x.so (the C ABI declared in include/plonky2_mi355x.h).

The library is HIP-only: importing this module works without a GPU (so that symbol/ABI checks can run on a
CPU box), but creating a context fails loudly when no MI355X is visible.  There is no CPU fallback and the
CPU oracle under oracle/ is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLONKY2_MI355X_LIB: another build of the same library (e.g. the -DNTT_ABLATION diagnostic build of tools/ablation.sh)
LIB_PATH = os.environ.get("PLONKY2_MI355X_LIB") or os.path.join(_HERE, "libplonky2_mi355x.so")

GL_OK = 0
GL_ERR_VERIFY = 6
ERRORS = {1: "GL_ERR_ARG", 2: "GL_ERR_HIP", 3: "GL_ERR_UNSUPPORTED", 4: "GL_ERR_ZETA_IN_SUBGROUP", 5: "GL_ERR_INTERNAL", 6: "GL_ERR_VERIFY"}


class Plonky2Mi355xError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s: %s" % (ERRORS.get(code, "error %d" % code), text))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libplonky2_mi355x.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C plonky2_demo_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    return ctypes.CDLL(LIB_PATH)


lib = _load()

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_u32 = ctypes.c_uint32
c_u64 = ctypes.c_uint64
c_int = ctypes.c_int

# name -> (restype, argtypes); every symbol include/plonky2_mi355x.h declares must appear here
SIGNATURES = {
    "gl_ctx_create": (c_int, [c_int, c_vp, ctypes.POINTER(c_vp)]),
    "gl_ctx_destroy": (None, [c_vp]),
    "gl_ctx_synchronize": (c_int, [c_vp]),
    "gl_ctx_set_scratch_elems": (c_int, [c_vp, c_sz]),
    "gl_last_error": (ctypes.c_char_p, []),
    "gl_ctx_timing_enable": (c_int, [c_vp, c_int]),
    "gl_ctx_timing_reset": (c_int, [c_vp]),
    "gl_ctx_timing_report": (c_int, [c_vp, ctypes.c_char_p, c_sz]),
    "gl_dev_alloc": (c_int, [c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "gl_dev_free": (c_int, [c_vp, c_vp]),
    "gl_copy_h2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "gl_copy_d2h": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "gl_field_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz]),
    "gl_ext_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_sz]),
    "gl_ntt_forward": (c_int, [c_vp, c_vp, c_u32, c_u32]),
    "gl_ntt_inverse": (c_int, [c_vp, c_vp, c_u32, c_u32]),
    "gl_ntt_coset_forward": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u64]),
    "gl_ntt_coset_inverse": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u64]),
    "gl_ntt_coset_lde": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "gl_fft_host": (c_int, [c_vp, c_vp, c_u32, c_u32, c_int]),
    "gl_poseidon_permute": (c_int, [c_vp, c_vp, c_sz]),
    "gl_hash_rows": (c_int, [c_vp, c_vp, c_sz, c_sz, c_vp]),
    "gl_merkle_new": (c_int, [c_vp, c_vp, c_sz, c_sz, c_u32, ctypes.POINTER(c_vp)]),
    "gl_merkle_cap": (c_int, [c_vp, c_vp]),
    "gl_merkle_prove": (c_int, [c_vp, c_sz, c_vp, ctypes.POINTER(c_u32)]),
    "gl_merkle_free": (None, [c_vp]),
    "gl_batch_from_values": (c_int, [c_vp, ctypes.POINTER(c_vp), c_sz, c_sz, c_u32, c_u32, c_u32, ctypes.POINTER(c_vp)]),
    "gl_batch_from_coeffs": (c_int, [c_vp, ctypes.POINTER(c_vp), c_sz, c_sz, c_u32, c_u32, c_u32, ctypes.POINTER(c_vp)]),
    "gl_batch_from_device": (c_int, [c_vp, c_vp, c_sz, c_sz, c_u32, c_u32, c_int, ctypes.POINTER(c_vp)]),
    "gl_batch_cap": (c_int, [c_vp, c_vp]),
    "gl_batch_get_leaf": (c_int, [c_vp, c_sz, c_vp]),
    "gl_batch_get_lde_values": (c_int, [c_vp, c_sz, c_sz, c_vp]),
    "gl_batch_prove": (c_int, [c_vp, c_sz, c_vp, ctypes.POINTER(c_u32)]),
    "gl_batch_coeffs": (c_int, [c_vp, c_vp]),
    "gl_batch_lde": (c_int, [c_vp, c_vp]),
    "gl_batch_ncols": (c_sz, [c_vp]),
    "gl_batch_degree": (c_sz, [c_vp]),
    "gl_batch_dev_coeffs": (c_vp, [c_vp]),
    "gl_batch_dev_lde": (c_vp, [c_vp]),
    "gl_batch_free": (None, [c_vp]),
    "gl_matmul_circuit_build": (c_int, [c_sz, ctypes.POINTER(c_vp)]),
    "gl_host_circuit_desc": (c_int, [c_vp, c_vp]),
    "gl_host_circuit_row_gates": (c_int, [c_vp, c_vp]),
    "gl_host_circuit_constants_sigmas": (c_int, [c_vp, c_vp]),
    "gl_matmul_witness": (c_int, [c_vp, c_vp, c_vp, c_u64, c_vp, c_vp]),
    "gl_host_circuit_free": (None, [c_vp]),
    "gl_matmul_witgen_create": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_matmul_witgen_run": (c_int, [c_vp, c_vp, c_vp, c_u64, c_vp, c_vp, c_vp]),
    "gl_matmul_witgen_free": (None, [c_vp]),
    "gl_circuit_create": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_circuit_from_host": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_circuit_description": (c_int, [c_vp, c_vp]),
    "gl_circuit_warm_up": (c_int, [c_vp, c_vp]),
    "gl_prover_pool_create_generic": (c_int, [c_int, c_vp, c_vp, c_u32, ctypes.POINTER(c_vp)]),
    "gl_prover_pool_prove_columns": (c_int, [c_vp, c_sz, c_vp, c_vp, c_vp]),
    "gl_circuit_digest": (c_int, [c_vp, c_vp]),
    "gl_circuit_constants_sigmas_cap": (c_int, [c_vp, c_vp]),
    "gl_circuit_constants_sigmas_batch": (c_vp, [c_vp]),
    "gl_circuit_free": (None, [c_vp]),
    "gl_partial_products": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_quotient_polys": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_partial_products_lookups": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_quotient_polys_lookups": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_open_at": (c_int, [c_vp, c_vp, c_vp, c_sz, c_sz, c_vp]),
    "gl_fri_combine": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "gl_fri_commit_round": (c_int, [c_vp, c_vp]),
    "gl_fri_fold": (c_int, [c_vp, c_vp]),
    "gl_fri_final_poly": (c_int, [c_vp, c_vp, c_sz, ctypes.POINTER(c_sz)]),
    "gl_pow_grind": (c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_vp]),
    "gl_fri_query": (c_int, [c_vp, c_vp, c_u32, c_vp, c_sz, ctypes.POINTER(c_sz)]),
    "gl_fri_free": (None, [c_vp]),
    "gl_ctx_capture_intermediates": (c_int, [c_vp, c_int]),
    "gl_challenger_new": (c_vp, []),
    "gl_challenger_observe": (c_int, [c_vp, c_vp, c_sz]),
    "gl_challenger_get_challenges": (c_int, [c_vp, c_vp, c_sz]),
    "gl_challenger_state": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(c_u32)]),
    "gl_challenger_free": (None, [c_vp]),
    "gl_prove": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "gl_prove_columns": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "gl_prove_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "gl_prove_device_hashed": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, ctypes.POINTER(c_vp)]),
    "gl_prover_pool_create": (c_int, [c_int, c_vp, c_u32, ctypes.POINTER(c_vp)]),
    "gl_prover_pool_lanes": (c_u32, [c_vp]),
    "gl_prover_pool_circuit": (c_vp, [c_vp]),
    "gl_prover_pool_prove_matmul": (c_int, [c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "gl_prover_pool_free": (None, [c_vp]),
    "gl_proof_num_bytes": (c_sz, [c_vp]),
    "gl_proof_bytes": (c_int, [c_vp, c_vp, c_sz]),
    "gl_proof_challenges": (c_sz, [c_vp, c_vp]),
    "gl_proof_caps": (c_int, [c_vp, c_vp]),
    "gl_proof_zs_partial_products": (c_int, [c_vp, c_vp]),
    "gl_proof_quotient_chunks": (c_int, [c_vp, c_vp]),
    "gl_proof_query_indices": (c_sz, [c_vp, c_vp]),
    "gl_proof_free": (None, [c_vp]),
    "gl_verify": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz]),
    "gl_host_circuit_verify": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz]),
    "gl_circuit_create_from_classes": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gl_host_circuit_wire_classes": (c_int, [c_vp, c_vp]),
    "gl_common_data_to_bytes": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "gl_common_data_from_bytes": (c_int, [c_vp, c_sz, c_vp, c_vp]),
    "gl_verifier_only_to_bytes": (c_int, [c_u32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "gl_verifier_only_from_bytes": (c_int, [c_vp, c_sz, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "gl_verify_bytes": (c_int, [c_vp, c_sz, c_vp, c_sz]),
}


class CircuitDesc(ctypes.Structure):
    """gl_circuit_desc (include/plonky2_mi355x.h)."""
    _fields_ = [
        ("degree_bits", c_u32), ("num_wires", c_u32), ("num_routed_wires", c_u32), ("num_constants", c_u32),
        ("num_selectors", c_u32), ("num_challenges", c_u32), ("quotient_degree_factor", c_u32),
        ("rate_bits", c_u32), ("cap_height", c_u32), ("proof_of_work_bits", c_u32), ("num_query_rounds", c_u32),
        ("num_fri_rounds", c_u32), ("fri_arity_bits", c_u32 * 8), ("num_public_inputs", c_u32), ("num_gates", c_u32),
        ("gate_types", ctypes.c_uint8 * 16), ("gate_params", ctypes.c_uint8 * 16), ("gate_selector_index", c_u32 * 16),
        ("gate_group_start", c_u32 * 16), ("gate_group_end", c_u32 * 16), ("k_is", c_u64 * 80),
        ("num_lookup_polys", c_u32), ("num_lookup_selectors", c_u32), ("num_luts", c_u32),
        ("last_lu_row", c_u32 * 4), ("last_lut_row", c_u32 * 4), ("first_lut_row", c_u32 * 4), ("lut_len", c_u32 * 4),
        ("lut", ctypes.c_uint16 * 2048),
    ]

    def lookup_table(self, t):
        """Table t as a list of (input, output) pairs: `lut` holds the tables one after the other."""
        off = sum(self.lut_len[i] for i in range(t))
        return [(self.lut[2 * (off + k)], self.lut[2 * (off + k) + 1]) for k in range(self.lut_len[t])]

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def check(status):
    if status != GL_OK:
        raise Plonky2Mi355xError(status, (lib.gl_last_error() or b"").decode())
