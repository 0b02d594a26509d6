"""Host compositions of BASELINE.json configurations 4 and 5 over the C-ABI entry points, with the
rank sharding of SURVEY.md 8(e): independent units (ciphertexts / matrix row blocks) are split into contiguous
blocks per rank, every rank runs the same launches on its block, no collective on the data path -- so the
results are bit-identical for every number of ranks.

* config 4: relinearize + Galois rotate of a batch of ciphertexts -- the flows of
  example_bfv_hybrid_key_switching / example_bfv_rotate_row (examples/1_bfv.cu:1269-1336, 1041-1157) at the
  parameter set of benchmark/keyswitch_bench.cu:25-34; relinearize_inplace src/evaluate.cu:1028-1077,
  apply_galois_inplace :1567-1624.
* config 5: encrypted matrix-vector product in diagonal form from hoisted rotations.  The reference has no such
  function (its matmul_bench is a plaintext modular GEMM); the composition is this build's, from the reference's
  hoisting_inplace (src/evaluate.cu:1670-1866), multiply_plain_inplace (:1297-1340) and add_inplace (:116-198).
"""
import torch

from . import dist as pdist
from .core import scheme_type


def relinearize_rotate_batch(ctx, size_Ql, ct3, relin_key, galois_key, galois_elt, scheme, chunk=0):
    """ct3 [B][3][Ql][N] (a batch of size-3 ciphertexts) -> [B][2][Ql][N]: relinearize, then rotate by
    galois_elt.  BFV ciphertexts are in coefficient form, CKKS / BGV in NTT form, as in the reference.
    One library call (pha_relinearize_rotate_batched): the batch goes through the batched key switches `chunk` ciphertexts
    at a time (0: sized so that the mod-up digits of a set stay within the 256 MiB MALL -- 8 at N = 2^15 / 30 + 15 limbs, where
    that measured fastest), with no copy between the stages."""
    B = ct3.shape[0]
    out = torch.empty_like(ct3[:, :2])
    if B:
        ctx.relinearize_rotate_batched(size_Ql, ct3.contiguous(), B, relin_key.public_keys_ptr, galois_key.public_keys_ptr,
                                       galois_elt, scheme, out, chunk)
    return out


def relinearize_rotate_batch_host(ctx, size_Ql, ct3, relin_key, galois_key, galois_elt, scheme, chunk=8):
    """The same from the two batched key switches and the Galois helper, composed on the host (the r02 form: three
    polynomial copies per ciphertext more); kept as the cross-check of the one-call form."""
    B = ct3.shape[0]
    out = torch.empty_like(ct3[:, :2])
    for b0 in range(0, B, max(1, chunk)):
        out[b0:b0 + chunk] = _relinearize_rotate_chunk(ctx, size_Ql, ct3[b0:b0 + chunk], relin_key, galois_key, galois_elt, scheme)
    return out


def _relinearize_rotate_chunk(ctx, size_Ql, ct3, relin_key, galois_key, galois_elt, scheme):
    B = ct3.shape[0]
    ct = ct3[:, :2].clone(memory_format=torch.contiguous_format)   # never a view: the key switch works in place
    ctx.keyswitch_inplace_batched(size_Ql, ct, ct3[:, 2].contiguous(), B, relin_key.public_keys_ptr, scheme)
    # rotate: (galois(c0), 0) += key switch of galois(c1) (apply_galois_inplace, src/evaluate.cu:1567-1624); one kernel
    # writes both operands in the layout the key switch wants
    rot = torch.empty_like(ct)
    g1 = torch.empty_like(ct[:, 0])
    ctx.apply_galois_for_keyswitch(ct, rot, g1, galois_elt, size_Ql, B, int(scheme) != int(scheme_type.bfv))
    ctx.keyswitch_inplace_batched(size_Ql, rot, g1, B, galois_key.public_keys_ptr, scheme)
    return rot


def relinearize_rotate_sharded(ctx, size_Ql, ct3, relin_key, galois_key, galois_elt, scheme, rank=None, world=None):
    """Config 4 on this rank: the contiguous block shard_range(B, rank, world) of the batch; returns
    (index range, results of that block)."""
    rank = pdist.rank() if rank is None else rank
    world = pdist.world() if world is None else world
    mine = pdist.shard_range(ct3.shape[0], rank, world)
    if len(mine) == 0:
        return mine, ct3[:0, :2].contiguous()
    return mine, relinearize_rotate_batch(ctx, size_Ql, ct3[mine.start:mine.stop], relin_key, galois_key, galois_elt, scheme)


def diag_matvec(ctx, size_Ql, ct, galois_elts, galois_keys, diagonals, scheme):
    """One block of config 5: out = sum_k diagonals[k] (.) rotate_{galois_elts[k]}(ct) (Halevi-Shoup diagonal
    form), all rotations hoisted behind one mod-up and one mod-down (pha_hoisting_weighted).  diagonals[k] is the
    k-th generalised diagonal encoded over [Q_l || P] in NTT form, [Ql + size_P][N]; Galois element 1 (the main
    diagonal) takes no key."""
    out = ct.clone()
    ctx.hoisting_weighted(size_Ql, out, galois_elts, galois_keys, diagonals, scheme)
    return out


def diag_matvec_bsgs(ctx, size_Ql, ct, baby_elts, baby_keys, giant_elts, giant_keys, diagonals, scheme):
    """The same block in baby-step / giant-step form (pha_hoisting_weighted_bsgs): with d = n_giant * n_baby diagonals,
    out = sum_i rot_{giant_elts[i]}(sum_j diagonals[i][j] (.) rot_{baby_elts[j]}(ct)) from n_baby + n_giant - 2 Galois keys instead of
    d - 1.  diagonals[i][j] is diagonal i * n_baby + j of the matrix, rotated back by giant step i and encoded over [Q_l || P]
    ([Ql + size_P][N], NTT form) -- the usual offline preparation of the BSGS matrix-vector product."""
    out = ct.clone()
    ctx.hoisting_weighted_bsgs(size_Ql, out, baby_elts, baby_keys, giant_elts, giant_keys, diagonals, scheme)
    return out


def diag_matvec_bsgs_blocks(ctx, size_Ql, ct, baby_elts, baby_keys, giant_elts, giant_keys, blocks, scheme, per_call=0):
    """Several row blocks of the matrix against ONE encrypted vector (pha_hoisting_weighted_bsgs_blocks): blocks[r][i][j] as in
    diag_matvec_bsgs.  `per_call` blocks go through one call (0: as many as make 16 (block, giant step) accumulators, the width of
    the fused baby-step kernel, so that the baby keys are streamed once per that many blocks).  Returns [len(blocks)][2][Ql][N];
    every block equals its own diag_matvec_bsgs."""
    import torch
    nblk = len(blocks)
    out = torch.empty((nblk,) + tuple(ct.shape), dtype=ct.dtype, device=ct.device)
    if per_call <= 0:
        per_call = max(1, 16 // max(1, len(giant_elts)))
    for r0 in range(0, nblk, per_call):
        ctx.hoisting_weighted_bsgs_blocks(size_Ql, ct, baby_elts, baby_keys, giant_elts, giant_keys, blocks[r0:r0 + per_call],
                                          out[r0:r0 + per_call], scheme)
    return out


def matvec_row_blocks_sharded(ctx, size_Ql, ct, galois_elts, galois_keys, blocks, scheme, rank=None, world=None):
    """Config 5 on this rank: `blocks` is a list of row blocks of the matrix, each a list of encoded diagonals
    (one output ciphertext per block); blocks are split contiguously over the ranks.  Returns (index range,
    list of output ciphertexts of that range)."""
    rank = pdist.rank() if rank is None else rank
    world = pdist.world() if world is None else world
    mine = pdist.shard_range(len(blocks), rank, world)
    return mine, [diag_matvec(ctx, size_Ql, ct, galois_elts, galois_keys, blocks[i], scheme) for i in mine]
