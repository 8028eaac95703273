"""Host-side composition of the HIP kernels (thin: pointer/stride bookkeeping only).

Every function here launches kernels from libfd_hip.so through se3_diffusion_amd.hip;
torch is used for memory (torch.empty / views) and nothing else.
"""
from __future__ import annotations

import math
import threading

import torch

from . import hip
from .options import opts

F32 = torch.float32


def lib():
    return hip.get_lib()


def empty(shape, like, dtype=F32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def zeros(shape, like, dtype=F32):
    a = _ARENA["buf"]
    if a is not None and dtype == F32 and a.device == like.device:
        n = 1
        for d in shape:
            n *= int(d)
        lo = _ARENA["used"]
        hi = lo + (n + 63) // 64 * 64                 # 256-byte aligned slices (16-byte operand paths of the kernels)
        if hi <= a.numel():
            _ARENA["used"] = hi
            return a[lo:lo + n].view(shape)
    return torch.zeros(shape, device=like.device, dtype=dtype)


# Zero arena: the backward pass needs ~15 zero-initialised accumulators per block (gradient sums that several launches add
# into).  One memset over one buffer replaces the ~60 fill launches of a step (7.5 us each); zeros() carves from it
# while it is open and falls back to torch.zeros when it is exhausted or closed.
_ARENA = {"buf": None, "used": 0}


class zero_arena:
    def __init__(self, numel, like):
        self.numel, self.like = int(numel), like

    def __enter__(self):
        self.prev = dict(_ARENA)
        _ARENA["buf"] = torch.zeros(self.numel, device=self.like.device, dtype=F32) if self.numel > 0 else None
        _ARENA["used"] = 0
        return self

    def __exit__(self, *exc):
        _ARENA.update(self.prev)
        return False


# ---------------------------------------------------------------------------
# dense helpers.  A "matrix view" is (tensor, elem_offset, ld): rows at offset + r*ld.
# ---------------------------------------------------------------------------
def mv(t, off=0, ld=None):
    return (t, off, ld if ld is not None else t.shape[-1])


# ---------------------------------------------------------------------------
# pre-split weights.  In training the parameters change once per step; when they live in ONE flat fp32 buffer (optim.FlatAdam,
# dist.FlatGrads) trunk.forward splits that buffer into its three exact bf16 planes with one launch (fd_split_planes) and every
# node-level GEMM of the forward and of the backward reads its weight operand from the planes (fd_gemm tiles 12-14: no split
# work and no register staging for the weights inside the kernel).  _PLANES = (first byte, bytes, [3, n] int16 planes) of the
# buffer that is currently split, or None: a weight outside it takes the old tiles.
# ---------------------------------------------------------------------------
_PLANES = None


def weight_planes(P):
    """Split the flat buffer that holds every tensor of P (a state-dict-like mapping) and make it the current plane set.  Returns the
    plane set (for trunk.backward, which re-installs it), or None when the parameters are not views of one 16-byte aligned buffer,
    the option is off or the library is in exact-fp32 mode."""
    global _PLANES
    _PLANES = None
    if not opts.weight_planes or lib().exact_f32:
        return None
    ts = list(P.values())
    st = ts[0].untyped_storage()
    base, nbytes = st.data_ptr(), st.nbytes()
    if base % 16 or nbytes % 32 or any(t.dtype != F32 or t.untyped_storage().data_ptr() != base for t in ts):
        return None
    n = nbytes // 4
    flat = torch.empty(0, dtype=F32, device=ts[0].device).set_(st, 0, (n,), (1,))
    planes = torch.empty((3, n), dtype=torch.int16, device=ts[0].device)
    lib().call("fd_split_planes", flat, n, planes)
    _PLANES = (base, nbytes, planes)
    return _PLANES


def set_weight_planes(ps):
    """Install (or clear, ps = None) a plane set returned by weight_planes(); returns the previous one."""
    global _PLANES
    was, _PLANES = _PLANES, ps
    return was


def _planes_of(wt, wo):
    """b_planes argument of FdLib.gemm for the weight view (wt, element offset wo), or None"""
    ps = _PLANES
    if ps is None:
        return None
    a = wt.data_ptr() + 4 * wo
    if not (ps[0] <= a < ps[0] + ps[1]):
        return None
    return (ps[2].data_ptr() + (a - ps[0]) // 2, ps[1] // 4)


def linear(x, W, b, out, M, N, K, *, relu=False, resid=None, rowscale=None, pair=None, beta=False,
           gate=None, alpha=1.0, tile=0):
    """out[M,N] = epi(x[M,K] @ W[N,K]^T + b).  x, W, out, resid, gate are matrix views."""
    xt, xo, xl = x
    wt, wo, wl = W
    ot, oo, ol = out
    kw = {}
    bp = _planes_of(wt, wo)
    if bp is not None:
        kw["b_planes"] = bp
    if resid is not None:
        kw.update(resid=(resid[0], resid[1]), ld_resid=resid[2])
    if gate is not None:
        kw.update(gate=(gate[0], gate[1]), ld_gate=gate[2])
    lib().gemm(xt, wt, ot, M, N, K, (xl, 1), (1, wl), ol, a_off=xo, b_off=wo, c_off=oo, bias=b,
               relu=relu, rowscale=rowscale, pair=pair, beta=beta, alpha=alpha, tile=tile, **kw)


def linear_dx(dy, W, dx, M, N, K, *, beta=False, gate=None, rowscale=None, alpha=1.0, resid=None):
    """dx[M,K] (+)= dy[M,N] @ W[N,K]; optional relu gate (zero where gate<=0) on the result; resid adds a
    second matrix view (dx = resid + dy W: a residual branch without accumulating in place)."""
    dt, do, dl = dy
    wt, wo, wl = W
    xt, xo, xl = dx
    kw = {}
    if gate is not None:
        kw.update(gate=(gate[0], gate[1]), ld_gate=gate[2])
    if resid is not None:
        kw.update(resid=(resid[0], resid[1]), ld_resid=resid[2])
    bp = _planes_of(wt, wo)
    if beta and not kw and rowscale is None and N >= 1024 and opts.dx_splitk and not lib().exact_f32:
        # an accumulating dX with a long reduction and few output tiles (IPA projections: 3840 x 256 over N = 2048 / 4096
        # is 240 tiles of 64 x 64 walking 64..128 stages each): split the reduction, the partial tiles add atomically
        # into the accumulator that is already there (order-nondeterministic: off in exact-fp32 mode, whose contract is a
        # bitwise reproducible fmaf chain)
        lib().gemm(dt, wt, xt, M, K, N, (dl, 1), (wl, 1), xl, a_off=do, b_off=wo, c_off=xo, alpha=alpha,
                   ksplit=min(8, N // 512), b_planes=bp)
        return
    lib().gemm(dt, wt, xt, M, K, N, (dl, 1), (wl, 1), xl, a_off=do, b_off=wo, c_off=xo, beta=beta,
               rowscale=rowscale, alpha=alpha, b_planes=bp, **kw)


# ---------------------------------------------------------------------------
# weight-gradient side stream.  Weight gradients (dW = dY^T X) do not feed anything until the optimiser, so they
# run on a second HIP stream beside the dX chain: the ~80 node-level ones per step (15-40 us, at most ~1 block per CU
# each) fill idle CUs, the pair-level ones (MFMA-bound) overlap the HBM-bound LayerNorm / reduction kernels of the
# main stream (40.5 -> 39.6 ms/step; 41.8 without the side stream).  options.opts.grad_stream = False disables it,
# opts.grad_stream_max_rows caps the row count of the launches that may move.  Contract with the callers: the operands handed to side() are never
# written again on the main stream (no in-place reuse), and they are kept alive until join_grad_stream().
# ---------------------------------------------------------------------------
_SIDE = {"streams": {}, "pending": [], "used": False}


def _new_side_stream(device):
    """The gradient side stream.  opts.side_stream_priority: HIP stream priority (lower number = higher priority; the value
    is clamped to the device's range) -- a LOW priority lets the hardware dispatcher prefer the waves of the main stream's
    (critical-path) kernels whenever both queues have work."""
    return torch.cuda.Stream(device=device, priority=int(opts.side_stream_priority))


def side(fn, tensors, rows):
    """Run fn() (weight-gradient launches reading `tensors`) on the gradient side stream."""
    t = tensors[0]
    if not (opts.grad_stream and t.is_cuda and rows <= opts.grad_stream_max_rows):
        fn()
        return
    key = t.device.index
    st = _SIDE["streams"].get(key)
    if st is None:
        st = _SIDE["streams"][key] = _new_side_stream(t.device)
    st.wait_stream(torch.cuda.current_stream())      # operands (and the zero-filled gradient buffers) are ready
    with torch.cuda.stream(st):
        fn()
    _SIDE["pending"].extend(tensors)
    _SIDE["used"] = True


def side_active(t, rows):
    """Would side(fn, (t, ...), rows) move fn to the gradient side stream?"""
    return bool(opts.grad_stream and t.is_cuda and rows <= opts.grad_stream_max_rows)


def grad_stream(device):
    """The gradient side stream of `device` (None when disabled or not a GPU): work queued on it after side_sync()
    runs behind every gradient launch issued so far on either stream."""
    if not (opts.grad_stream and device.type == "cuda"):
        return None
    st = _SIDE["streams"].get(device.index)
    if st is None:
        st = _SIDE["streams"][device.index] = _new_side_stream(device)
    st.wait_stream(torch.cuda.current_stream())
    _SIDE["used"] = True
    return st


def set_grad_stream(on):
    """Enable / disable the gradient side stream; returns the previous setting."""
    was = opts.grad_stream
    opts.grad_stream = bool(on)
    return was


def join_grad_stream():
    """Main stream waits for every side-stream gradient launch; releases the operand references."""
    flush_dw()
    if _SIDE["used"]:
        cur = torch.cuda.current_stream()
        for st in _SIDE["streams"].values():
            cur.wait_stream(st)
        _SIDE["pending"].clear()
        _SIDE["used"] = False


# ---------------------------------------------------------------------------
# fork / join for the sampling forward.  A lone backbone's forward is ~146 dependent launches of 5-12 us; a handful of them
# do not depend on their neighbours (skip_embed of a block, the IPA point rotation beside q k^T, a v_pts + o_pt beside a v,
# the backbone update beside the edge transition).  fork(fn, like) runs fn() on a second stream that first waits for the
# current one; join(like) makes the current stream wait for it.  Under hipGraph capture (sampler.sample, use_graph) the
# pair becomes two edges of the graph and the branch runs beside the main chain.  A Python closure keeps variable CELLS, not
# values: a caller that rebinds a name the branch reads (trunk.forward: node, quat, trans = n3, q2, t2) would let the old tensor
# go back to the allocator while the side stream may still be reading it -- so callers bind what the branch reads as default
# arguments of fn and/or pass it as keep=(...); both stay referenced until the join.  Same single-caller contract as side().
# ---------------------------------------------------------------------------
_FORK = {"streams": {}, "pending": []}


def fork_ok(like):
    return bool(opts.graph_fork and like.is_cuda and lib().is_device)


def fork(fn, like, enable=True, keep=()):
    if not (enable and fork_ok(like)):
        fn()
        return
    key = like.device.index
    st = _FORK["streams"].get(key)
    if st is None:
        st = _FORK["streams"][key] = torch.cuda.Stream(device=like.device)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    _FORK["pending"].append((key, (fn, tuple(keep))))


def join(like=None):
    if _FORK["pending"]:
        cur = torch.cuda.current_stream()
        for key in {k for k, _ in _FORK["pending"]}:
            cur.wait_stream(_FORK["streams"][key])
        _FORK["pending"].clear()


# smallest K_in that sends an N_out = 128 pair-row weight gradient to the 128-row split-bf16 tile (256: the edge
# transition's; 96 would add the edge embedder's 128 x 128 / 128 x 120 layers, measured slower)
_DW_T6_MIN_K = 256


def linear_dw(dy, x, dW, M, N, K, db=None):
    """dW[N,K] += dy[M,N]^T @ x[M,K]  (reduction over the M rows; split-K); db[N] += sum_m dy[m,:] fused
    into the same kernel (row sums of the A = dy^T operand)."""
    dt, do, dl = dy
    xt, xo, xl = x
    wt, wo, wl = dW
    # output is tiny (N x K weights), the reduction (M rows) is huge: split K so that ~2300 blocks exist
    al16 = all((t.data_ptr() + 4 * o) % 16 == 0 and ld % 4 == 0 for t, o, ld in (dy, x))
    if N >= 384 and K >= 128 and M >= 65536 and N % 4 == 0 and K % 4 == 0 and al16:
        # large pair-level weight gradients (>= 75 % of the 256 x 128 tiles used, enough tiles that the split-K
        # atomics stay cheap): split-bf16 kernel with the fused row sum
        tile = 4
        blocks = ((N + 255) // 256) * ((K + 127) // 128)
    elif N == 128 and K >= _DW_T6_MIN_K and M >= 65536 and K % 4 == 0 and al16:
        # N_out = 128: the 128-row shape of the split kernel (0.44 vs 0.52 ms on the 64x64 fp32 tile at K_in = 384)
        tile = 6
        blocks = (K + 127) // 128
    else:
        tile, t = (1, 128) if (N >= 256 and K >= 256 and db is None) else (2, 64)
        blocks = ((N + t - 1) // t) * ((K + t - 1) // t)
    # ~9 blocks per CU for the 64/128 tiles; the 256x128 split-bf16 tile runs one block per CU, and every extra split
    # costs a full tile of L2 atomics: 3 per CU
    ks = max(1, min((768 if tile in (4, 6) else 2304) // max(1, blocks), M // 256))
    lib().gemm(dt, xt, wt, N, K, M, (1, dl), (xl, 1), wl, a_off=do, b_off=xo, c_off=wo,
               beta=(ks == 1), ksplit=ks, tile=tile, a_rowsum=db)


# ---------------------------------------------------------------------------
# grouped node-level weight gradients (csrc/fd_group_dw.hip).  The backward pass QUEUES every per-residue dW = dY^T X
# (+ bias gradient) it meets; flush_dw() executes the queue as one launch on the gradient side stream -- once per trunk
# block instead of ~25 split-K GEMM launches.  Same contract as side(): queued operands are never written again and stay
# referenced until join_grad_stream().
# ---------------------------------------------------------------------------
_DWQ = {"items": [], "tensors": [], "rows": None}
GROUP_DW_MAX_ROWS = 16384          # residue rows (B*N <= 5e5 / N); pair-row gradients have their own kernels


def queue_dw(dy, x, dW, M, N, K, db=None):
    """Queue dW[N,K] += dy[M,N]^T @ x[M,K] (db[N] += column sums of dy) for the next flush_dw().  Returns False when the
    product does not qualify (option off, exact-fp32 mode, operands not float4-addressable): the caller launches it itself."""
    if not queue_dw_ok(dy, x, dW, M, N, K):
        return False
    dt, xt, wt = dy[0], x[0], dW[0]
    q = _DWQ
    if q["rows"] is not None and q["rows"] != M:
        flush_dw()
    q["rows"] = M
    q["items"].append((dy, x, dW, db, int(N), int(K)))
    q["tensors"].extend(t for t in (dt, xt, wt, db) if t is not None)
    if len(q["items"]) == hip.GROUP_DW_MAX_ITEMS:
        flush_dw()
    return True


def group_dw(items, rows, blocks=0):
    """One fd_group_dw launch on the CURRENT stream.  items: (dy view, x view, dW view, db tensor | None, n_out, k_in) with views
    = (tensor, element offset, row stride):  dW[n_out, k_in] += dy^T x,  db += column sums of dy."""
    assert 1 <= len(items) <= hip.GROUP_DW_MAX_ITEMS
    d = hip.FdGroupDwDesc()
    tens = []
    for t, (dy, x, dW, db, n, k) in enumerate(items):
        e = d.item[t]
        e.A, e.B, e.C = hip._ptr(dy[0], dy[1]), hip._ptr(x[0], x[1]), hip._ptr(dW[0], dW[1])
        e.a_colsum = None if db is None else hip._ptr(db)
        e.lda, e.ldb, e.ldc, e.n_out, e.k_in = int(dy[2]), int(x[2]), int(dW[2]), int(n), int(k)
        tens.extend(t_ for t_ in (dy[0], x[0], dW[0], db) if t_ is not None)
    d.nitems, d.rows, d.blocks = len(items), int(rows), int(blocks or opts.node_dw_blocks)
    L = lib()
    stream = L._stream(tens)
    prof = L.gemm_profile
    if prof is not None and L.is_device:
        # profile record (tile code 11): 2 * rows * n_out * k_in flops per item
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        L._check(L.cdll.fd_group_dw(hip.ctypes.byref(d), stream), "fd_group_dw")
        e1.record(torch.cuda.current_stream())
        prof.append((11, False, False, sum(2.0 * rows * it[4] * it[5] for it in items), e0, e1,
                     (len(items), 128, int(rows), 1, 0, 0, 0, 1)))
        return
    L._check(L.cdll.fd_group_dw(hip.ctypes.byref(d), stream), "fd_group_dw")


def reset_dw_queue():
    """Drop whatever an interrupted backward pass left queued (an exception between queue_dw() and the next flush_dw()): a stale
    item would otherwise be launched by the NEXT step's first flush and add into that step's gradient buffers."""
    q = _DWQ
    q["items"], q["tensors"], q["rows"] = [], [], None


def queue_dw_ok(dy, x, dW, M, N, K):
    """Would queue_dw accept this product?  (callers that queue several products as a unit check all of them first)"""
    if not opts.grouped_node_dw or M > GROUP_DW_MAX_ROWS or M < 1 or lib().exact_f32:
        return False
    dt, do, dl = dy
    xt, xo, xl = x
    wt, wo, wl = dW
    if (N % 4) or (K % 4) or (dl % 4) or (xl % 4) or (dt.data_ptr() + 4 * do) % 16 or (xt.data_ptr() + 4 * xo) % 16:
        return False
    return not (wl < K or dl < N or xl < K)


def flush_dw(blocks=0):
    """Launch the queued weight gradients (one fd_group_dw per <= 32 items) on the gradient side stream."""
    q = _DWQ
    if not q["items"]:
        return
    items, tens, rows = q["items"], q["tensors"], q["rows"]
    q["items"], q["tensors"], q["rows"] = [], [], None
    side(lambda: group_dw(items, rows, blocks), tens, rows)


def add_view(dst, src, rows, cols, alpha=1.0):
    """dst(view)[r, c] += alpha * src(view)[r, c]."""
    dt, do, dl = dst
    st, so, sl = src
    lib().call("fd_add2d", (dt, do), dl, (st, so), sl, rows, cols, alpha)


def bias_grad(dy, db, M, N):
    dt, do, dl = dy
    lib().call("fd_colsum_acc", (dt, do), dl, M, N, db)


def layernorm(x, gamma, beta, y, rows, C, *, rowscale=None, save=None):
    """y = LN(x) (* rowscale).  x, y matrix views.  save=(mean, rstd) tensors or None."""
    xt, xo, xl = x
    yt, yo, yl = y
    mean, rstd = save if save is not None else (None, None)
    lib().call("fd_layernorm_fwd", (xt, xo), xl, gamma, beta, rowscale, (yt, yo), yl, mean, rstd, rows, C, 1e-5)


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, rows, C, *, rowscale=None, dgamma=None, dbeta=None, accum=False):
    dt, do, dl = dy
    xt, xo, xl = x
    gt, go, gl = dx
    lib().call("fd_layernorm_bwd", (dt, do), dl, (xt, xo), xl, gamma, rowscale, mean, rstd, (gt, go), gl,
               int(accum), dgamma, dbeta, rows, C)


def ln_linear_ok(x, W, M, N, K):
    """Can ln_linear fold this LayerNorm into its Linear?  (the latency regime of sampling, M = B*N <= 1024 rows -- where fd_gemm
    picks its 32 x 32 latency kernel for the node-level layers --, K <= 320, contiguous 16-byte aligned operands)"""
    return bool(opts.ln_fold and M <= 1024 and K % 8 == 0 and K <= 320 and x[1] % 4 == 0 and W[1] == 0 and x[2] % 4 == 0
                and W[2] % 4 == 0)


def ln_linear(x, gamma, beta, W, b, out, M, N, K, *, relu=False, resid=None, ln_rowscale=None, ln_out=None, ln_cols=0):
    """out[M,N] = epi(LN(x)[M,K] @ W[N,K]^T + b) with LN(x) = ln_rowscale * (LayerNorm(x) * gamma + beta) formed inside the GEMM
    launch (csrc/fd_ln_gemm.hip); ln_out (a matrix view): LN(x) written out too; ln_cols: only x[:, :ln_cols] is normalised.  x, W, out, resid are matrix views."""
    d = hip.FdLnGemmDesc()
    tens = []

    def ptr(t, off=0):
        if t is None:
            return None
        tens.append(t)
        return t.data_ptr() + 4 * off
    d.x, d.ldx = ptr(x[0], x[1]), x[2]
    d.gamma, d.beta, d.ln_rowscale = ptr(gamma), ptr(beta), ptr(ln_rowscale)
    if ln_out is not None:
        d.ln_out, d.ld_ln_out = ptr(ln_out[0], ln_out[1]), ln_out[2]
    d.W, d.ldw, d.bias = ptr(W[0], W[1]), W[2], ptr(b)
    if resid is not None:
        d.resid, d.ld_resid = ptr(resid[0], resid[1]), resid[2]
    d.out, d.ldo = ptr(out[0], out[1]), out[2]
    d.M, d.N, d.K, d.relu, d.eps, d.ln_cols = int(M), int(N), int(K), int(bool(relu)), 1e-5, int(ln_cols)
    L = lib()
    L._check(L.cdll.fd_ln_gemm(hip.ctypes.byref(d), L._stream(tens)), "fd_ln_gemm")


# ---------------------------------------------------------------------------
# host-computed constant tables (reference op sequence, so arguments are bit-identical)
# ---------------------------------------------------------------------------
_TABLES = {}


def feature_tables(device, index_embed_size=32, num_bins=22, min_bin=1e-5, max_bin=20.0):
    key = (str(device), index_embed_size, num_bins, min_bin, max_bin)
    if key not in _TABLES:
        half = index_embed_size // 2
        # score_network.py:40-42
        tfreq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
        # score_network.py:26-29: max_len ** (2*K/embed_size)
        k = torch.arange(half)
        idenom = (2056 ** (2 * k[None] / index_embed_size))[0].to(torch.float32)
        # data/utils.py:573-578
        lower = torch.linspace(min_bin, max_bin, num_bins)
        upper = torch.cat([lower[1:], lower.new_tensor([1e8])])
        _TABLES[key] = tuple(t.contiguous().to(device) for t in (tfreq, idenom, lower, upper))
    return _TABLES[key]


# ---------------------------------------------------------------------------
# fused edge transition (csrc/fd_edge_mlp.hip)
# ---------------------------------------------------------------------------
def edge_mlp_pack(W1, W2, Wf, backward=False, out=None, W40=None):
    """Pack the edge-transition weights (trunk.0 [384,384], trunk.2 [384,384], final_layer [128,384]) into the bf16-plane
    image fd_edge_mlp streams.  backward=True packs the transposed chain (dX kernel).  W40 [40,128]: forward -- the next IPA
    block's [linear_b ; down_z] as a fourth layer (edge_mlp(..., zb_out=, zb_bias=)); backward -- the IPA block BEHIND the
    transition, for the dzb term of the fused prologue (edge_mlp(..., ln_y=, dzb=))."""
    img = out if out is not None else torch.empty(hip.EDGE_MLP_IMAGE_BYTES, dtype=torch.uint8, device=W1.device)
    assert W40 is None or (W40.is_contiguous() and tuple(W40.shape) == (40, 128))
    assert W1.stride(-1) == 1 and W2.stride(-1) == 1 and Wf.stride(-1) == 1 and W1.stride(0) == W2.stride(0) == Wf.stride(0)
    ld = W1.stride(0)
    if not backward:
        lib().call("fd_edge_mlp_pack", W1, W2, Wf, ld, img)
        if W40 is not None:
            lib().call("fd_edge_mlp_pack_zb", W40, img)
    else:
        lib().call("fd_edge_mlp_pack_bwd", Wf, W2, W1, ld, W40, img)
    return img


def edge_mlp_pack_bwd(Wf, W2, W1, W40=None, out=None):
    """Backward image (edge_mlp(..., backward=True)), preceded by W40^T [128 <- 40] (the IPA block behind the transition) when
    the dzb term of the fused prologue is used."""
    return edge_mlp_pack(W1, W2, Wf, backward=True, out=out, W40=W40)


_SCHED = {}
_SCHED_LOCK = threading.Lock()
# launches that took the dynamic tile hand-out since the process started (tests assert that the benchmarked step uses it)
STATS = {"edge_dynamic_launches": 0, "ipa_flash_fwd": 0, "ipa_sequence_fwd": 0, "ipa_flash_bwd": 0, "ipa_sequence_bwd": 0,
         "ipa_flash_bwd_keys": 0, "ipa_keys_gemms": 0}


_SCHED_WORDS = 256           # counter words per device: one per (stream, launch site) that ever used the dynamic hand-out


def edge_sched_init(device):
    """Allocate the device's pool of tile-counter words.  Called outside any hipGraph capture (ScoreNetwork.forward's first
    call on a device, sampler.sample before it captures): a first dynamic-tile launch ON a capture stream (sampling at N=512)
    must not allocate inside the capture -- the buffer would live in the graph's private pool while this module keeps it."""
    with _SCHED_LOCK:
        pool = _SCHED.get(("pool", device))
        if pool is None:
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ops.edge_sched_init must run before hipGraph capture starts (sampler.sample calls it)")
            pool = _SCHED[("pool", device)] = (torch.zeros(_SCHED_WORDS, dtype=torch.int32, device=device), {})
    return pool


def _edge_sched(like, stream):
    """One word of device scratch for a launch's dynamic tile hand-out (FdEdgeMlpDesc.sched; the entry point zeroes it on the
    launch's stream).  One word per (device, stream), taken from a per-device pool that exists before any capture: two launches
    on one stream are ordered, so the second launch's zeroing memset runs after the first has ended; launches on different
    streams (two models, a replayed graph beside eager work) never share a word.  Captured graphs keep the word of their
    capture stream (replays of ONE graph are ordered by the stream they are replayed on; replaying two graphs captured on the
    same stream concurrently is not supported)."""
    buf, slots = edge_sched_init(like.device)
    with _SCHED_LOCK:
        key = int(stream or 0)
        slot = slots.get(key)
        if slot is None:
            if len(slots) >= _SCHED_WORDS // 4:
                # (never wrap: two streams sharing a word would skip / duplicate tiles of concurrent launches without an error)
                raise RuntimeError(f"ops._edge_sched: more than {_SCHED_WORDS // 4} streams have used the dynamic tile hand-out on "
                                   f"{like.device}; raise ops._SCHED_WORDS")
            slot = slots[key] = len(slots)                           # (4-word stride: 16-byte aligned words)
        STATS["edge_dynamic_launches"] += 1
    return buf.data_ptr() + 16 * slot


EDGE_MLP_MACS_PER_ROW = 128 * 384 + 384 * 384 + 384 * 128      # 245,760 (rounds 2-4 also ran Wf[:, :128] z: 262,144)


def edge_mlp(x, img, out, rows, nres, *, p1=None, q1=None, bias2=None, pf=None, qf=None, gamma=None, beta=None,
             rowscale=None, save1=None, save2=None, y=None, mean=None, rstd=None,
             backward=False, blocks=0, ld_pq=0, ld_pqf=0, zb_out=None, zb_bias=None, mask1=None, mask2=None, gmask1=None,
             gmask2=None, ln_y=None, ln_mean=None, ln_rstd=None, ln_gamma=None, ln_rowscale=None, dy_out=None, ln_dgamma=None,
             ln_dbeta=None, dzb=None):
    d = hip.FdEdgeMlpDesc()
    tens = []
    for name, t in (("x", x), ("img", img), ("p1", p1), ("q1", q1), ("bias2", bias2),
                    ("save1", save1), ("save2", save2), ("pf", pf), ("qf", qf), ("gamma", gamma), ("beta", beta),
                    ("rowscale", rowscale), ("y", y), ("mean", mean), ("rstd", rstd), ("out", out), ("zb_out", zb_out),
                    ("zb_bias", zb_bias), ("mask1", mask1), ("mask2", mask2), ("gmask1", gmask1), ("gmask2", gmask2),
                    ("ln_y", ln_y), ("ln_mean", ln_mean), ("ln_rstd", ln_rstd), ("ln_gamma", ln_gamma), ("ln_rowscale", ln_rowscale),
                    ("dy_out", dy_out), ("ln_dgamma", ln_dgamma), ("ln_dbeta", ln_dbeta), ("dzb", dzb)):
        setattr(d, name, None if t is None else t.data_ptr())
        if t is not None:
            tens.append(t)
    d.rows, d.nres, d.backward, d.eps, d.blocks = int(rows), int(nres), int(bool(backward)), 1e-5, int(blocks or opts.edge_blocks)
    d.ld_pq, d.ld_pqf = int(ld_pq), int(ld_pqf)
    d.shape = int(opts.edge_shape)
    # shape 2 (two waves per 16-row group, csrc/fd_edge_mlp_pair.hip) exists for the inference forward only: a forced 2 applies to
    # those launches and leaves the others to the size rule; edge_pair=False keeps the size rule from picking it
    infer_fwd = not backward and save1 is None
    if d.shape == 2 and not infer_fwd:
        d.shape = 0
    if d.shape == 0 and infer_fwd and not opts.edge_pair and rows <= hip.EDGE_MLP_PAIR_MAX_ROWS:
        d.shape = 4
    L = lib()
    stream = L._stream(tens)
    # tiles of the shape the entry point will pick: 128 rows on 256 blocks (8 waves, >= EDGE_MLP_W8_MIN_ROWS rows) or 64 rows on 512
    w8 = d.shape == 8 or (d.shape == 0 and rows >= hip.EDGE_MLP_W8_MIN_ROWS)
    if opts.edge_dynamic_tiles and rows >= 4 * (128 if w8 else 64) * (d.blocks or (256 if w8 else 512)):
        d.sched = _edge_sched(out, stream)
    prof = L.gemm_profile
    if prof is not None and L.is_device:
        # same record layout as FdLib.gemm (tile code 7 = the fused edge-transition kernel); algorithmic flops of the
        # chain: 2 * rows * (128*384 + 384*384 + 384*128) -- the residual through the final layer is an add, not a product
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        L._check(L.cdll.fd_edge_mlp(hip.ctypes.byref(d), stream), "fd_edge_mlp")
        e1.record()
        prof.append((7, True, True, 2.0 * int(rows) * (EDGE_MLP_MACS_PER_ROW + (5120 if zb_out is not None else 0)), e0, e1,
                     (int(rows), 128, 384, 1, int(bool(backward)), 0, 0, 1)))
        return
    L._check(L.cdll.fd_edge_mlp(hip.ctypes.byref(d), stream), "fd_edge_mlp")


# ---------------------------------------------------------------------------
# fused edge embedder (csrc/fd_edge_embed.hip)
# ---------------------------------------------------------------------------
def edge_embed_pack(W0, W2, W4, out=None, W40=None):
    img = out if out is not None else torch.empty(hip.EDGE_EMBED_IMAGE_BYTES, dtype=torch.uint8, device=W0.device)
    lib().call("fd_edge_embed_pack", W0, W2, W4, img)
    if W40 is not None:
        assert W40.is_contiguous() and tuple(W40.shape) == (40, 128)
        lib().call("fd_edge_embed_pack_zb", W40, img)
    return img


def edge_embed(seq_idx, sc_ca, idenom, dg_lower, dg_upper, img, p, q, bias2, bias3, gamma, beta, out, rows, nres, *,
               rowscale=None, h1=None, h2=None, h3=None, mean=None, rstd=None, blocks=0, zb_out=None, zb_bias=None, mask1=None,
               mask2=None, ld_pq=0):
    d = hip.FdEdgeEmbedDesc()
    tens = []
    for name, t in (("seq_idx", seq_idx), ("sc_ca", sc_ca), ("idenom", idenom), ("dg_lower", dg_lower), ("dg_upper", dg_upper),
                    ("img", img), ("p", p), ("q", q), ("bias2", bias2), ("bias3", bias3), ("gamma", gamma), ("beta", beta),
                    ("rowscale", rowscale), ("h1", h1), ("h2", h2), ("h3", h3), ("mean", mean), ("rstd", rstd), ("out", out),
                    ("zb_out", zb_out), ("zb_bias", zb_bias), ("mask1", mask1), ("mask2", mask2)):
        setattr(d, name, None if t is None else t.data_ptr())
        if t is not None:
            tens.append(t)
    d.rows, d.nres, d.eps, d.blocks = int(rows), int(nres), 1e-5, int(blocks)
    d.ld_pq = int(ld_pq)
    L = lib()
    stream = L._stream(tens)
    prof = L.gemm_profile
    if prof is not None and L.is_device:
        # profile record (tile code 8): algorithmic flops of the per-pair part: 2 * rows * (54 + 128 + 128) * 128
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        L._check(L.cdll.fd_edge_embed(hip.ctypes.byref(d), stream), "fd_edge_embed")
        e1.record()
        prof.append((8, True, True, 2.0 * int(rows) * 310 * 128, e0, e1, (int(rows), 128, 128, 1, 0, 0, 0, 1)))
        return
    L._check(L.cdll.fd_edge_embed(hip.ctypes.byref(d), stream), "fd_edge_embed")


def edge_embed_bwd_pack(W2, W4, out=None):
    img = out if out is not None else torch.empty(hip.EDGE_EMBED_BWD_IMAGE_BYTES, dtype=torch.uint8, device=W2.device)
    lib().call("fd_edge_embed_bwd_pack", W2, W4, img)
    return img


def edge_embed_bwd(dy, h3, mean, rstd, gamma, rowscale, h2, h1, img, dh3, dh2, dh1, dgamma, dbeta, rows, blocks=0, gmask2=None,
                   gmask1=None):
    """The edge embedder's LayerNorm backward + dX chain in one launch (csrc/fd_edge_embed_bwd.hip)."""
    d = hip.FdEdgeEmbedBwdDesc()
    tens = []
    for name, t in (("dy", dy), ("h3", h3), ("mean", mean), ("rstd", rstd), ("gamma", gamma), ("rowscale", rowscale), ("h2", h2),
                    ("h1", h1), ("img", img), ("dh3", dh3), ("dh2", dh2), ("dh1", dh1), ("dgamma", dgamma), ("dbeta", dbeta),
                    ("gmask2", gmask2), ("gmask1", gmask1)):
        setattr(d, name, None if t is None else t.data_ptr())
        if t is not None:
            tens.append(t)
    d.rows, d.blocks = int(rows), int(blocks)
    L = lib()
    stream = L._stream(tens)
    if opts.edge_dynamic_tiles and rows >= 4 * 64 * (d.blocks or 512):
        d.sched = _edge_sched(dh3, stream)
    L._check(L.cdll.fd_edge_embed_bwd(hip.ctypes.byref(d), stream), "fd_edge_embed_bwd")


# ---------------------------------------------------------------------------
# grouped pair-row weight gradients (csrc/fd_pair_dw.hip)
# ---------------------------------------------------------------------------
def pair_dw(items, rows, blocks=0):
    """One launch for up to 8 tiles  C[m, n] += sum_p (A[p, m] + [m < 128] A_add[p, m]) B[p, n]  (m < 384, n < 128).

    items: dicts with A=(tensor, offset, ld) [rows,384], B=(tensor, offset, ld) [rows,128], C=(tensor, offset, ld) and
    optionally A_add=(tensor, offset, ld) [rows,128], colsum=tensor [384], trans=bool (C[n, m]); a_bands=1: A is
    [rows,128] (a 128 x 128 tile); b_cols=k: B is [rows,k], k % 4 == 0 (C has k columns)."""
    d = hip.FdPairDwDesc()
    tens = []
    assert 1 <= len(items) <= hip.PAIR_DW_MAX_ITEMS
    for t, it in enumerate(items):
        e = d.item[t]
        for name, ld in (("A", "lda"), ("B", "ldb"), ("C", "ldc"), ("A_add", "ld_add")):
            v = it.get(name)
            if v is None:
                setattr(e, name, None)
                setattr(e, ld, 0)
                continue
            ten, off, stride = v
            setattr(e, name, hip._ptr(ten, off))
            setattr(e, ld, int(stride))
            tens.append(ten)
        cs = it.get("colsum")
        e.a_colsum = None if cs is None else hip._ptr(cs)
        e.trans = int(bool(it.get("trans", False)))
        e.a_bands = int(it.get("a_bands", 3))
        e.b_cols = int(it.get("b_cols", 0))
    d.nitems, d.rows, d.blocks = len(items), int(rows), int(blocks)
    L = lib()
    stream = L._stream(tens)
    prof = L.gemm_profile
    if prof is not None and L.is_device:
        # profile record (tile code 9): 2 * rows * 384 * 128 flops per item
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        L._check(L.cdll.fd_pair_dw(hip.ctypes.byref(d), stream), "fd_pair_dw")
        e1.record(torch.cuda.current_stream())
        flops = sum(2.0 * int(rows) * 128 * int(it.get("a_bands", 3)) * (int(it.get("b_cols", 0)) or 128) for it in items)
        prof.append((9, False, False, flops, e0, e1, (384, 128 * len(items), int(rows), 1, 0, 0, 0, 1)))
        return
    L._check(L.cdll.fd_pair_dw(hip.ctypes.byref(d), stream), "fd_pair_dw")
