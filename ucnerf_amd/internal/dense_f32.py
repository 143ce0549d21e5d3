"""fp32 dense layers of the NON-autocast training step on hand-written kernels (csrc/gemm_h3.hip, r06; csrc/gemm_f32.hip, r04).

The reference's shipped launch trains in fp32 (scripts/train_waymo.sh:3 has no --mixed_precision, train.py:165's autocast is then
a no-op): every nn.Linear of the NeRF / proposal fields (internal/models.py:438-483, 507-674), of the sky NeRF (models.py:743-820) and
of the colour-correction head (internal/extrinsic_optimizer.py:4-48) is an fp32 `F.linear` forward and two fp32 GEMMs backward.
`hip_linear(x, W, b)` is that triple on `ucn_gemm_f32` / `ucn_wgrad_f32` (exact fp32 products on v_mfma_f32_32x32x2_f32, fp32
accumulation): forward Y = X W^T + b (ReLU fused on request), d X = d Y W (the same kernel on the transposed weight), d W = d Y^T X and
d b = column sums of d Y in one pass (fixed-order partial sums: deterministic).

The kernels take 16-byte operand loads: reduction lengths and leading dimensions are multiples of 4 floats.  The few operands of this
model that are not (3 / 27 / 283-wide inputs, 1 / 3-wide outputs) are zero-padded copies made here -- they are per-ray or weight-sized,
never an activation-sized concatenation.

r06: the default engine is "split" -- the same three functions on csrc/gemm_h3.hip: fp32-class products x w ~ x_hi w_hi + x_hi w_lo +
x_lo w_hi of f16 halves on v_mfma_f32_32x32x16_f16 (16 / 3 of the fp32 MFMA rate), fp32 accumulation, every operand at a power-of-two
scale taken from its absolute maximum.  That maximum is a DEVICE float that travels with the tensor (`_ucn_amax`, set by the GEMM whose
epilogue produced the tensor; computed by one ucn_amax_f32 pass for operands that come from elsewhere) -- no host round trip.  The
exact fp32-product kernels stay behind `set_engine("exact")` / UCN_F32_EXACT=1; tall operands only (M >= 32768), short ones always run
exact.

UCN_F32_LIBRARY=1 (experiment switch, read per call) routes the same functions through torch's library GEMMs: the A/B measurement of
bench.py `train_step_fp32`, not a fallback -- on a host tensor these functions raise like every other entry point.
"""
import os

import torch

from .. import _lib

ACCUMULATE, RELU, MASK = 1, 2, 4

H3_MIN_ROWS = 32768             # below this a GEMM is launch-bound either way (per-ray terms, the colour head): exact products, no pack
_ENGINE = "exact" if os.environ.get("UCN_F32_EXACT", "0") == "1" else "split"


def engine():
    return _ENGINE


def set_engine(name):
    """"split": fp32-class split-f16 products (csrc/gemm_h3.hip; default).  "exact": exact fp32 products (csrc/gemm_f32.hip)."""
    global _ENGINE
    assert name in ("split", "exact"), name
    prev, _ENGINE = _ENGINE, name
    return prev


def library_route():
    return os.environ.get("UCN_F32_LIBRARY", "0") == "1"


# ---- absolute maxima of the split engine's operands: device floats that travel with their tensors ------------------------------------
_POOL = {}


def _slot(device):
    """a zeroed device float (a view into a pool that is zero-filled once per 4096 slots)"""
    key = str(device)
    hit = _POOL.get(key)
    if hit is None or hit[1] >= hit[0].numel():
        hit = _POOL[key] = [torch.zeros(4096, device=device), 0]
    s = hit[0][hit[1]:hit[1] + 1]
    hit[1] += 1
    return s


def _tag(t, slot):
    t._ucn_amax = (slot, t._version)
    return t


def forget(t):
    """call after writing into a tensor through its raw pointer (kernels do not bump `_version`): its recorded maximum is stale"""
    if hasattr(t, "_ucn_amax"):
        del t._ucn_amax
    if hasattr(t, "_ucn_relu_bits"):
        del t._ucn_relu_bits


def amax_of(t):
    """device float holding max |t| (an upper bound is as good) of a [M, K] operand: the one recorded when a kernel of this module
    produced `t`, else one ucn_amax_f32 pass (then remembered on the tensor object)."""
    hit = getattr(t, "_ucn_amax", None)
    if hit is not None and hit[1] == t._version:
        return hit[0]
    lib = _lib.load()
    s = _slot(t.device)
    M, K = t.shape
    _lib.check(lib.ucn_amax_f32(t.data_ptr(), t.stride(0), M, K, s.data_ptr(), _lib.stream()))
    _tag(t, s)
    return s


def tag_amax_of_parts(whole, *parts):
    """`whole` was assembled from `parts` (column blocks written by kernels / copies): its maximum is the largest of theirs -- one
    tiny device op instead of a pass over `whole`.  Parts without a recorded maximum get one (amax_of)."""
    if _ENGINE != "split" or not whole.is_cuda:
        return whole
    m = None
    for p_ in parts:
        s = amax_of(p_ if p_.dim() == 2 else p_.reshape(-1, p_.shape[-1]))
        m = s if m is None else torch.maximum(m, s)
    return _tag(whole, m)


def stash_amax(ctx, tensors):
    """autograd: remember the recorded maxima of tensors about to be saved for backward (saved_tensors may hand back new objects)"""
    ctx._ucn_amax = [getattr(t, "_ucn_amax", None) if t is not None else None for t in tensors]
    ctx._ucn_bits = [getattr(t, "_ucn_relu_bits", None) if t is not None else None for t in tensors]      # (and their ReLU bit masks)


def restore_amax(ctx, tensors):
    for t, a in zip(tensors, getattr(ctx, "_ucn_amax", ())):
        if t is not None and a is not None and a[1] == t._version:
            t._ucn_amax = a
    for t, b in zip(tensors, getattr(ctx, "_ucn_bits", ())):
        if t is not None and b is not None and b[1] == t._version and b[2] == t.data_ptr():
            t._ucn_relu_bits = b


RELU_BITS = os.environ.get("UCN_RELU_BITS", "1") != "0"     # A/B switch: 0 = the d X GEMMs read the stored fp32 outputs as their masks


def _bits_shape_ok(out, N, bias, rowbias, mask=None):
    """the split engine's staged epilogue (the one that writes / reads ReLU bit masks): 128- or 256-wide outputs, 16-byte rows"""
    return (RELU_BITS and N in (128, 256) and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0
            and (bias is None or bias.data_ptr() % 16 == 0)
            and (rowbias is None or (rowbias.stride(0) % 4 == 0 and rowbias.data_ptr() % 16 == 0))
            and (mask is None or (mask.stride(0) % 4 == 0 and mask.data_ptr() % 16 == 0)))


def _bits_of(mask, M, N):
    """the bit form of a ReLU mask: recorded by the split engine when it produced `mask` as a ReLU output of this very shape"""
    hit = getattr(mask, "_ucn_relu_bits", None)
    if hit is not None and hit[1] == mask._version and hit[2] == mask.data_ptr() and hit[3] == (M, N) and tuple(mask.shape) == (M, N):
        return hit[0]
    return None


def _rows(t):
    """[M, K] float32 device view with unit column stride, a row stride that is a multiple of 4 and a 16-byte aligned base; K padded
    to a multiple of 4 with zero columns if it is not (copy).  Views that already conform (column slices at multiples of 4 of a wider
    buffer) are passed as they are."""
    _lib.require_device(t, "dense_f32 operand")
    src = t
    if t.dtype != torch.float32:
        t = t.float()
    t = t.reshape(-1, t.shape[-1])
    k = t.shape[1]
    ok = (t.stride(1) == 1 or k == 1) and t.stride(0) % 4 == 0 and t.stride(0) >= k and t.data_ptr() % 16 == 0 and k % 4 == 0
    if ok:
        if t is not src and hasattr(src, "_ucn_amax") and t.data_ptr() == src.data_ptr() and t.numel() == src.numel():
            t._ucn_amax = src._ucn_amax                                  # the same values under another shape
        return t
    kp = (k + 3) // 4 * 4
    if kp == k:
        return t.contiguous()
    out = t.new_zeros(t.shape[0], kp)
    out[:, :k] = t
    return out


ACT_PAD = int(os.environ.get("UCN_ACT_PAD", "0"))          # experiment: floats added to the row stride of wide activation buffers


def rows_buffer(M, N, device):
    """an uninitialised [M, N] float32 activation buffer for `out=`.  UCN_ACT_PAD > 0 (experiment, measured SLOWER): on the split engine
    a column view of an [M, N + ACT_PAD] allocation, so that rows are not 1024 bytes apart (tools/h3_alias_probe.py had the 256 x 256
    product alone at 0.60 - 0.66 ms against 0.72 with such strides at M = 2^20; in the step, M = 983 040 / 1.8 M rows, the fp32 sky step
    went 32.7 -> 34.3 / 34.4 / 35.0 ms with 8 / 24 / 72 floats of padding and the literal launch 49.3 -> 51.6 / 51.9 / 52.3)."""
    if _ENGINE == "split" and ACT_PAD > 0 and M >= H3_MIN_ROWS and N % 128 == 0:
        return torch.empty(M, N + ACT_PAD, device=device, dtype=torch.float32)[:, :N]
    return torch.empty(M, N, device=device, dtype=torch.float32)


def gemm(x, w, bias=None, flags=0, out=None, n_out=None, mask=None, rowbias=None, rgroup=0, x2=None, w2=None):
    """out[M, N] (+)= x[M, K] w[N, K]^T (+ bias) (+ rowbias[row // rgroup]) (+ x2[M, 4] w2[N, 4]^T) (ReLU); x / w as returned by _rows
    (equal padded K).  `out` may be a column view.  mask [M, N] (a float tensor, e.g. the stored ReLU output of the layer below):
    out = mask > 0 ? out : 0 as the last step of the epilogue.  x2 / w2 (r06): a second, 4-wide operand pair (zero-padded 3-d points
    against their weight columns, a density-row gradient against its weight row) added inside the split engine's epilogue; on the exact
    engine and for short operands the same sum as a second accumulating call."""
    lib = _lib.load()
    M, K = x.shape
    if x2 is not None and not (_ENGINE == "split" and M >= H3_MIN_ROWS and (w.shape[0] if n_out is None else n_out) == 256):
        # two passes: the wide product (bias / row bias), then the narrow one accumulating, with the ReLU / mask behind the sum
        first = int(flags) & ACCUMULATE
        out = gemm(x, w, bias, first, out, n_out, None, rowbias, rgroup)
        return gemm(x2, w2, None, (int(flags) & ~ACCUMULATE) | ACCUMULATE, out, n_out, mask)
    N = w.shape[0] if n_out is None else n_out
    assert w.shape[1] == K, (x.shape, w.shape)
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    assert out.stride(1) == 1 or N == 1
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape[0] == M and mask.shape[1] >= N and (mask.stride(1) == 1 or N == 1)
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rgroup > 0 and rowbias.shape[0] * rgroup >= M and rowbias.shape[1] >= N and rowbias.stride(1) == 1
    if _ENGINE == "split" and M >= H3_MIN_ROWS and N <= 256:
        # fp32-class products on the split-f16 engine: pack the weight (its scale from its own maximum), the activations' scale from the
        # maximum recorded by the kernel that produced them, this product's maximum recorded for the next one
        packed = torch.empty(lib.ucn_pack_h3_bytes(N, K), device=x.device, dtype=torch.uint8)
        wmax = _slot(x.device)
        _lib.check(lib.ucn_pack_h3(w.data_ptr(), w.stride(0), N, K, 0, packed.data_ptr(), wmax.data_ptr(), _lib.stream()))
        xmax = amax_of(x)
        ymax = _slot(x.device)          # (ACCUMULATE too: the epilogue sees, and records, the final values of every element of `out`)
        if x2 is not None:
            assert x2.shape == (M, 4) and w2.shape[1] == 4 and w2.shape[0] >= N and x2.dtype == w2.dtype == torch.float32
            assert x2.stride(1) == 1 and w2.stride(1) == 1
        # ReLU derivatives as bits (r06): a ReLU product leaves "out > 0" as one bit per element next to its fp32 output; a masked
        # product whose mask carries such bits reads them instead of the fp32 tensor (4 bytes -> 1 bit per element of the d X epilogue)
        mbits = None
        if mask is not None and _bits_shape_ok(out, N, bias, rowbias):
            mbits = _bits_of(mask, M, N)
        if mbits is not None:
            mask = None
        forget(out)
        obits = None
        if (int(flags) & RELU) and _bits_shape_ok(out, N, bias, rowbias, mask):
            obits = torch.empty(lib.ucn_relu_bits_words(M, N), device=x.device, dtype=torch.int64)
        _lib.check(lib.ucn_gemm_h3_x2(x.data_ptr(), x.stride(0), packed.data_ptr(), xmax.data_ptr(), wmax.data_ptr(), _lib.ptr(bias), M, N, K,
                                      int(flags) | (MASK if mask is not None else 0), out.data_ptr(), out.stride(0), _lib.ptr(mask),
                                      0 if mask is None else mask.stride(0), _lib.ptr(rowbias), 0 if rowbias is None else rowbias.stride(0),
                                      int(rgroup), _lib.ptr(x2), 0 if x2 is None else x2.stride(0), _lib.ptr(w2),
                                      0 if w2 is None else w2.stride(0), _lib.ptr(obits), _lib.ptr(mbits), ymax.data_ptr(), _lib.stream()))
        if obits is not None:
            out._ucn_relu_bits = (obits, out._version, out.data_ptr(), (M, N))
        return _tag(out, ymax)
    forget(out)
    if mask is None and rowbias is None:
        _lib.check(lib.ucn_gemm_f32(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias), M, N, K, int(flags),
                                    out.data_ptr(), out.stride(0), _lib.stream()))
        return out
    if mask is not None:
        flags = int(flags) | MASK
    _lib.check(lib.ucn_gemm_f32_ex(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias), M, N, K, int(flags),
                                   out.data_ptr(), out.stride(0), _lib.ptr(mask), 0 if mask is None else mask.stride(0),
                                   _lib.ptr(rowbias), 0 if rowbias is None else rowbias.stride(0), int(rgroup), _lib.stream()))
    return out


_WS = {}


def wgrad(gy, x, want_bias=False):
    """(gy^T x as [N, K] float32, column sums of gy [N] or None); gy [M, N], x [M, K] as returned by _rows."""
    lib = _lib.load()
    M, N = gy.shape
    K = x.shape[1]
    if not want_bias and K <= 64 < N:
        # a wide gradient against a narrow input ([256, 64]: the composed colour layers): the kernel's blocks are 256 k columns wide,
        # its narrow shapes are narrow in N -- form the transpose (x^T gy: a 64-row block, every k column used) and flip it back
        return wgrad(x, gy, False)[0].t().contiguous(), None
    split = _ENGINE == "split" and M >= H3_MIN_ROWS
    n = (lib.ucn_wgrad_h3_ws_floats if split else lib.ucn_wgrad_f32_ws_floats)(N, K, M)
    st = torch.cuda.current_stream()
    key = (str(gy.device), st.cuda_stream)
    hit = _WS.get(key)
    if hit is None or hit[0] != st or hit[1].numel() < n:
        if len(_WS) > 8:
            _WS.clear()
        hit = _WS[key] = (st, torch.empty(max(n, 1), device=gy.device))
    gw = torch.empty(N, K, device=gy.device)
    gb = torch.empty(N, device=gy.device) if want_bias else None
    if split:
        _lib.check(lib.ucn_wgrad_h3(gy.data_ptr(), gy.stride(0), x.data_ptr(), x.stride(0), amax_of(gy).data_ptr(), amax_of(x).data_ptr(),
                                    M, N, K, hit[1].data_ptr(), gw.data_ptr(), _lib.ptr(gb), _lib.stream()))
        return gw, gb
    _lib.check(lib.ucn_wgrad_f32(gy.data_ptr(), gy.stride(0), x.data_ptr(), x.stride(0), M, N, K, hit[1].data_ptr(), gw.data_ptr(),
                                 _lib.ptr(gb), _lib.stream()))
    return gw, gb


class _HipLinear(torch.autograd.Function):
    """torch.nn.functional.linear(x, weight, bias) (+ ReLU) in fp32 on the hand-written kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        lead, K, N = x.shape[:-1], x.shape[-1], weight.shape[0]
        x2, w2 = _rows(x), _rows(weight)
        y = gemm(x2, w2, None if bias is None else bias.float().contiguous(), RELU if relu else 0)
        ctx.save_for_backward(x2, weight, y if relu else None)
        stash_amax(ctx, (x2,))
        ctx.meta = (lead, K, N, bias is not None, relu, x.dtype, weight.dtype)
        out = y.reshape(lead + (N,))
        if hasattr(y, "_ucn_amax"):
            out._ucn_amax = (y._ucn_amax[0], out._version)
        return out

    @staticmethod
    def backward(ctx, gy):
        x2, weight, y = ctx.saved_tensors
        restore_amax(ctx, (x2,))
        lead, K, N, has_bias, relu, x_dt, w_dt = ctx.meta
        gy2 = gy.reshape(-1, N)
        if relu:
            gy2 = torch.ops.aten.threshold_backward(gy2.contiguous(), y, 0.0)      # gy where y > 0 else 0: one elementwise pass
        g4 = _rows(gy2)                                                  # [M, N padded to 4]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = _rows(weight.detach().t())                              # [K, N padded]: d X = d Y W as the forward kernel on W^T
            gx = gemm(g4, wt)[:, :K].reshape(lead + (K,)).to(x_dt)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw_p, gb_p = wgrad(g4, x2, has_bias)
            gw = gw_p[:N, :K].to(w_dt)
            gb = None if gb_p is None else gb_p[:N]
        return gx, gw, gb, None


def hip_linear(x, weight, bias=None, relu=False):
    """F.linear(x, weight, bias) [+ ReLU] for float32 device tensors: csrc/gemm_f32.hip forward and backward."""
    if library_route():
        y = torch.nn.functional.linear(x, weight, bias)
        return torch.relu(y) if relu else y
    return _HipLinear.apply(x, weight, bias, relu)


def usable(*tensors):
    """the fp32 route's kernels take float32 device tensors outside autocast"""
    return (not torch.is_autocast_enabled()) and all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors)
