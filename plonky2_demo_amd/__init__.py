"""plonky2_demo_amd -- MI355X-native backend for the prove() hot path of the Plonky2 matmul demo circuit.

Host-side mirror (Python, over the C ABI of libplonky2_mi355x.so) of the reference interfaces on the path:
``fft`` / ``ifft`` (field/src/fft.rs:52-95), ``PolynomialBatch`` (plonky2/src/fri/oracle.rs:30-133),
``MerkleTree`` (plonky2/src/hash/merkle_tree.rs:39-207), ``poseidon`` (plonky2/src/hash/poseidon.rs:598-609).
Names and argument meaning follow the reference so that the parity tests read like its own tests.
"""
from ._lib import LIB_PATH, Plonky2Mi355xError  # noqa: F401
from . import api  # noqa: F401
from .api import (  # noqa: F401
    Challenger, CircuitData, Context, FriProver, GenericCircuitData, MatmulCircuit, MerkleTree, PolynomialBatch, Proof, ProverPool, coset_fft, coset_ifft, default_context, fft, hash_or_noop, ifft,
    lde_onto_coset, poseidon, pow_grind, GOLDILOCKS_ORDER, COSET_SHIFT,
)
