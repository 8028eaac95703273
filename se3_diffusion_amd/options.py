"""Launch-selection options of the hot path: ONE object, read at call time.

Every alternative launch sequence the path can take (a fused kernel against the launches it replaces, the gradient side
stream, the flat-layout fast paths) is selected by a field of ``opts``.  The fields are initialised once from ``FD_*``
environment variables (so ``tools/ablation.sh`` keeps working) and can be flipped afterwards -- by tests, without
re-importing anything:

    from se3_diffusion_amd import options
    with options.override(fused_edge=False, grad_stream=False):
        ...

Both settings of every field are covered by a parity test (tests/test_switches.py, tests/test_parity_full.py's exact-fp32
mode, tests/test_seq_attn.py).  ``FD_GEMM_EXACT_F32`` is not here: it is state of the loaded library
(``hip.FdLib.set_exact_f32``), because the C side consults it too.
"""
from __future__ import annotations

import contextlib
import dataclasses
import os


def _flag(name, default):
    return os.environ.get(name, "1" if default else "0") not in ("", "0")


def _int(name, default):
    return int(os.environ.get(name, str(default)))


@dataclasses.dataclass
class Options:
    # -- streams
    grad_stream: bool = True          # FD_GRAD_STREAM: weight gradients on a second HIP stream beside the dX chain (ops.side)
    grad_stream_max_rows: int = 1 << 40   # FD_GRAD_STREAM_ROWS: row count above which a launch stays on the main stream
    side_stream_priority: int = 0     # FD_SIDE_PRIORITY: HIP priority of that stream (lower = higher; clamped to the device's range)
    # -- pair level
    fused_edge: bool = True           # FD_EDGE_FUSED: fd_edge_mlp (whole edge-transition chain per pair row) vs the GEMM sequence
    fused_embed: bool = True          # FD_EMBED_FUSED: fd_edge_embed (features generated in registers) vs fd_edge_feats + GEMMs
    fused_embed_bwd: bool = True      # FD_EMBED_BWD_FUSED: the edge embedder's dX chain in one launch (fd_edge_embed_bwd)
    grouped_pair_dw: bool = True      # FD_PAIR_DW: the edge transition's pair-row weight gradients in one grouped launch
    pair_dw_blocks: int = 160         # FD_PAIR_DW_BLOCKS: blocks of fd_pair_dw when it runs beside the main stream (0 = 256)
    edge_blocks: int = 0              # FD_EDGE_BLOCKS: persistent blocks of the fused edge kernels (0 = fill the CUs: 512 / 256)
    edge_shape: int = 0               # FD_EDGE_SHAPE: 0 = by size, 4 = 4-wave blocks (two per CU), 8 = 8-wave blocks (one per CU),
                                      # 2 = two waves per 16-row group (inference forward launches only; the others keep the size rule)
    edge_pair: bool = True            # FD_EDGE_PAIR: sampling -- an edge transition of at most 16,384 pair rows (a lone N <= 128 backbone:
                                      # one 16-row group per SIMD) on the column-split kernel, two waves per group (fd_edge_mlp_pair.hip)
    edge_dynamic_tiles: bool = True   # FD_EDGE_DYN_TILES: a fused edge launch with more tiles than blocks hands them out dynamically
    packed_gates: bool = True         # FD_PACKED_GATES: the fused edge EMBEDDER's backward gates on packed sign bits instead of reading h1 / h2
                                      # (the edge transition's always does: its h2 save carries the residual z)
    fused_ln_bwd: bool = True         # FD_EDGE_LN_BWD: the edge transition's LayerNorm backward (and the IPA term dz += dzb W40 of
                                      # the block behind it) as the prologue of its fused backward kernel
    zb_from_edge: bool = True         # FD_ZB_FUSED: the next IPA block's pair projection zb as a 4th layer of fd_edge_mlp
    fold_node_terms: bool = True      # FD_FOLD_NODE_TERMS: sampling -- per-residue terms of an edge transition as one GEMM
    # -- IPA
    fused_ipa_attn: bool = True       # FD_IPA_ATTN_FUSED: logits + softmax + o_pair of a query row in one launch
    flash_ipa: bool = True            # FD_IPA_FLASH: q k^T, logits, softmax, a v, a v_pts, o_pt and o_pair of a block in ONE launch
                                      # (fd_ipa_flash_fwd: the probabilities never reach HBM in inference) ...
    flash_ipa_min_tiles: int = 80     # FD_IPA_FLASH_MIN_TILES: ... from this many 16-row query tiles (B * ceil(N / 16)) up: a lone
                                      # backbone has too few tiles to fill the CUs and keeps the launch sequence (measured: 95 tiles
                                      # 151 against 180 us with the probabilities written, 75 tiles 192 against 195, 56 tiles 210 against 171)
    flash_ipa_bwd: bool = True        # FD_IPA_FLASH_BWD: the query side of IPA's attention backward in one launch (fd_ipa_flash_bwd: no dA
                                      # in HBM) instead of two batched GEMMs + fd_ipa_attn_bwd's per-row kernel ...
    flash_ipa_keys: bool = True       # FD_IPA_FLASH_KEYS: ... and its key side (dV, dv_pts, dK, dk_pts) in one launch (fd_ipa_flash_bwd_keys: A and dL
                                      # read once) instead of three batched GEMMs + fd_ipa_kpts_bwd, where the query side runs fused ...
    flash_ipa_keys_max_n: int = 384   # FD_IPA_FLASH_KEYS_MAX_N: ... up to this N (four key tiles of a head per block, the query operands through
                                      # LDS once per block: B=30 x N=128 78 against 102 us, B=12 x N=200 97 against 119, B=7 x N=256 82 against
                                      # 109; B=8 x N=512 248 against 225, B=1 x N=512 138 against 102: long chains keep the batched GEMMs)
    flash_ipa_bwd_min_tiles: int = 128  # FD_IPA_FLASH_BWD_MIN_TILES: ... from this many query tiles up (8 heads per block only: 112 tiles
                                      # 195 against 199 us, 95 tiles 224 against 207)
    flash_ipa_split_min_n: int = 384  # FD_IPA_FLASH_SPLIT_MIN_N: inference below flash_ipa_min_tiles -- from this N up the KEYS of a query tile are
                                      # split over 4 blocks + a merge launch (fd_ipa_flash_fwd_split: 86 against 112 us at N=512 B=1; 55
                                      # against 49 at N=256, 42 against 39 at N=128: the launch sequence stays there)
    flash_ipa_splits: int = 4         # FD_IPA_FLASH_SPLITS: ... key splits of that form
    flash_ipa_hpb: int = 0            # FD_IPA_FLASH_HPB: heads per block of that kernel (0 = by size, 8 / 4 / 2)
    proj_merge: bool = True           # FD_PROJ_MERGE: IPA's four projections of s as one GEMM over back-to-back weights
    # -- node level
    fused_seq_attn: bool = True       # FD_SEQ_ATTN_FUSED: sequence-transformer attention in one launch ...
    seq_attn_min_rows: int = 1024     # FD_SEQ_ATTN_MIN_ROWS: ... from this many residue rows up
    fused_seq_attn_bwd: bool = True   # FD_SEQ_ATTN_BWD_FUSED: its backward (dQ, dK, dV from the saved probabilities and output) in one launch
                                      # (fd_seq_attn_bwd) instead of four batched GEMMs + the row-softmax backward
    grouped_node_dw: bool = True      # FD_NODE_DW: the node-level weight gradients of a trunk block in one grouped launch
    defer_node_dw: bool = True        # FD_DEFER_NODE_DW: ... launched behind the NEXT edge transition's fused backward (beside that block's
                                      # node-level phase) instead of at the end of its own block (in front of that full-chip kernel)
    node_dw_blocks: int = 0           # FD_NODE_DW_BLOCKS: its persistent blocks (0 = 512: two per CU)
    weight_planes: bool = True        # FD_WEIGHT_PLANES: training -- the flat parameter buffer split into its three bf16 planes once per step
                                      # (fd_split_planes); the node-level GEMMs then read the weight operand pre-split (fd_gemm tiles 12-14)
    ln_fold: bool = True              # FD_LN_FOLD: sampling -- the sequence transformer's LayerNorms inside the GEMM launches that
                                      # consume them (fd_ln_gemm) instead of launches of their own
    sampler_device_steps: bool = True # FD_SAMPLER_DEVICE_STEPS: sampling -- the captured step takes t, the step's scalars and its normal draws
                                      # from device arrays indexed by a device counter (fd_sample_advance) instead of three launches per step
    embed_first_padded: bool = True   # FD_EMBED_FIRST_PADDED: sampling -- the per-residue features at a row stride of 72 (fd_node_feats_ld) so that the
                                      # embedders' first layers (K = 65 / 33) run on the latency GEMM; p | q of the edge embedder as ONE product
    merge_skip_embed: bool = True     # FD_MERGE_SKIP: sampling -- the skip_embed products of all trunk blocks as ONE GEMM per forward
    graph_fork: bool = False          # FD_GRAPH_FORK (measured, OFF: 1.33-1.40 against 1.61 backbones/s at N=128, 1.06-1.09 against 1.23 at
                                      # N=256 -- a cross-queue edge of the hipGraph costs ~20 us, more than the 5-12 us launch it hides): sampling -- launches that do not depend on each other (skip_embed, the IPA point
                                      # rotation beside q k^T, a v_pts + o_pt beside a v, the backbone update beside the edge transition) go to a
                                      # second stream and join before their consumer: parallel branches of the captured hipGraph
    # -- backward bookkeeping
    zero_arena: bool = True           # FD_ZERO_ARENA: one memset for every zero-initialised accumulator of a backward pass
    dx_splitk: bool = True            # FD_DX_SPLITK: accumulating dX GEMMs with a long reduction split over K (atomics)

    @classmethod
    def from_env(cls):
        return cls(
            grad_stream=_flag("FD_GRAD_STREAM", True), grad_stream_max_rows=_int("FD_GRAD_STREAM_ROWS", 1 << 40),
            side_stream_priority=_int("FD_SIDE_PRIORITY", 0),
            fused_edge=_flag("FD_EDGE_FUSED", True), fused_embed=_flag("FD_EMBED_FUSED", True),
            fused_embed_bwd=_flag("FD_EMBED_BWD_FUSED", True),
            grouped_pair_dw=_flag("FD_PAIR_DW", True), pair_dw_blocks=_int("FD_PAIR_DW_BLOCKS", 160),
            edge_blocks=_int("FD_EDGE_BLOCKS", 0), edge_shape=_int("FD_EDGE_SHAPE", 0), edge_pair=_flag("FD_EDGE_PAIR", True), embed_first_padded=_flag("FD_EMBED_FIRST_PADDED", True), zb_from_edge=_flag("FD_ZB_FUSED", True),
            fused_ln_bwd=_flag("FD_EDGE_LN_BWD", True), edge_dynamic_tiles=_flag("FD_EDGE_DYN_TILES", True), packed_gates=_flag("FD_PACKED_GATES", True),
            fold_node_terms=_flag("FD_FOLD_NODE_TERMS", True),
            fused_ipa_attn=_flag("FD_IPA_ATTN_FUSED", True),
            flash_ipa=_flag("FD_IPA_FLASH", True), flash_ipa_min_tiles=_int("FD_IPA_FLASH_MIN_TILES", 80),
            flash_ipa_bwd_min_tiles=_int("FD_IPA_FLASH_BWD_MIN_TILES", 128),
            flash_ipa_split_min_n=_int("FD_IPA_FLASH_SPLIT_MIN_N", 384),
            flash_ipa_hpb=_int("FD_IPA_FLASH_HPB", 0), flash_ipa_splits=_int("FD_IPA_FLASH_SPLITS", 4), flash_ipa_bwd=_flag("FD_IPA_FLASH_BWD", True), flash_ipa_keys=_flag("FD_IPA_FLASH_KEYS", True), flash_ipa_keys_max_n=_int("FD_IPA_FLASH_KEYS_MAX_N", 384),
            proj_merge=_flag("FD_PROJ_MERGE", True),
            fused_seq_attn=_flag("FD_SEQ_ATTN_FUSED", True), seq_attn_min_rows=_int("FD_SEQ_ATTN_MIN_ROWS", 1024), fused_seq_attn_bwd=_flag("FD_SEQ_ATTN_BWD_FUSED", True),
            grouped_node_dw=_flag("FD_NODE_DW", True), defer_node_dw=_flag("FD_DEFER_NODE_DW", True), node_dw_blocks=_int("FD_NODE_DW_BLOCKS", 0),
            ln_fold=_flag("FD_LN_FOLD", True), weight_planes=_flag("FD_WEIGHT_PLANES", True),
            sampler_device_steps=_flag("FD_SAMPLER_DEVICE_STEPS", True), merge_skip_embed=_flag("FD_MERGE_SKIP", True), graph_fork=_flag("FD_GRAPH_FORK", False),
            zero_arena=_flag("FD_ZERO_ARENA", True), dx_splitk=_flag("FD_DX_SPLITK", True))


opts = Options.from_env()


@contextlib.contextmanager
def override(**kw):
    """Temporarily set fields of ``opts`` (unknown names raise)."""
    was = {}
    for k, v in kw.items():
        if not hasattr(opts, k):
            raise AttributeError(f"unknown option {k!r}")
        was[k] = getattr(opts, k)
        setattr(opts, k, v)
    try:
        yield opts
    finally:
        for k, v in was.items():
            setattr(opts, k, v)
