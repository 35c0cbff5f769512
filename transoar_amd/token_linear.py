"""nn.Linear over the flattened feature pyramid (10^5 tokens per volume).

Forward and input gradient are plain NT GEMMs (x (T, K) . W (N, K)^T with the bias in the epilogue; dX = dY . W is
the same product with W^T as the (N, K)-shaped operand).  Two implementations, chosen per shape by measured time at
the 234 000-token shapes (profiles/r02_gemm_bench.jsonl): the hand-written bf16 MFMA GEMM of csrc/gemm.hip for the
K = N = 384 products -- value_proj, output_proj, the stacked sampling_offsets|attention_weights projection and
their data gradients: 0.15-0.16 ms = hipBLASLt's 0.16 ms -- and hipBLASLt for the FFN shapes 384 -> 1024 -> 384
(0.27 / 0.22 ms against 0.37 / 0.27 ms).  TRANSOAR_HIP_GEMM=1 / 0 forces one of them everywhere.

The weight gradient
dW = dY^T X contracts over the TOKEN axis (K = 234 000 at batch 2) into a tile
of at most 1024 x 384: hipBLASLt covers that with a few dozen workgroups on a
256-CU part (0.5-0.9 ms, 40-230 TFLOP/s measured).  Here the token axis is cut
into chunks that run as one batched GEMM with fp32 partial outputs, summed
afterwards -- 3-5x faster (tools/probe_wgrad.py), and the partial sums stay in
fp32 exactly like the single GEMM's accumulator.

Same arithmetic contract as autocast's nn.Linear: bf16 operands, fp32
accumulation, bf16 output; parameters and their gradients stay fp32.
"""
import os

import torch
import torch.nn.functional as F

from . import conv_gemm, gemm, rows, shadow

MIN_TOKENS = 32768          # below this the stock path is as fast
LAST_PATH = None            # "hip-gemm" / "blas": which forward ran last (tests)
OWN_FFN2 = os.environ.get("TRANSOAR_OWN_FFN2", "1") != "0"
USE_HIP_GEMM = {"1": True, "0": False}.get(os.environ.get("TRANSOAR_HIP_GEMM", ""), None)     # None: per shape


def _hip_gemm(x2, w, force=False):
    """Hand-written kernel for this product?  (x2 (T, K), w (N, K))"""
    if not gemm.usable(x2, w):
        return False
    if USE_HIP_GEMM is not None:
        return USE_HIP_GEMM
    # K = N = 384 on the tiled kernel (round 2); every K = 384 or N = 384 product of >= 16 384 tokens on the streaming
    # kernels of csrc/gemm_stream.hip (round 4: the FFN shapes 384 -> 1024 -> 384 and their data gradients included)
    # ... and 1024 -> 384 (linear2 and linear1's data gradient) on the tiled kernel as well: 0.27 ms against hipBLASLt's
    # 0.22, the price (0.1 ms per step) of a refinement block without a library GEMM.  Long K only: the Swin stages'
    # 96 -> 384 products run the dynamic-K loop with two K steps and cost the Swin step 14 ms when they came here
    return (force or (x2.shape[1] == 384 and w.shape[0] == 384) or gemm.stream_kind(x2, w) is not None
            or (OWN_FFN2 and w.shape[0] == 384 and x2.shape[0] >= 16384 and x2.shape[1] >= 512 and x2.shape[1] % 128 == 0))


def _chunks(tokens, n_out, n_in):
    tiles = -(-n_out // 64) * -(-n_in // 128)
    want = max(8, min(80, 1440 // max(tiles, 1)))
    size = max(1024, tokens // want)
    return tokens // size, size


# The hand-written voxel-major ("TN") GEMM of csrc/conv_gemm.hip (its one-tap case).  At the 234 000-token shapes
# (tools/bench_gemm.py, profiles/r03_gemm_bench.jsonl) it is within 5-13 % of round 2's chunked hipBLASLt batch on
# 384 x 384 / 384 -> 1024 / 1024 -> 384 (0.16 / 0.31 / 0.31 ms against 0.15 / 0.29 / 0.28) and ahead on the stacked
# 384 -> 576 projection and the FPN's 384 -> 3072 one (0.25 / 0.79 against 0.28 / 0.88): it is the default, so that no
# library GEMM is left in the weight-gradient path; TRANSOAR_HIP_WGRAD=0 goes back to the batch.
USE_HIP_WGRAD = os.environ.get("TRANSOAR_HIP_WGRAD", "1") == "1"


def weight_grad(gy, x):
    """(T, N)^T @ (T, K) -> (N, K) fp32: the token axis is the contraction."""
    t, n = gy.shape
    k = x.shape[1]
    if USE_HIP_WGRAD and gemm.wgrad384_usable(gy, x):
        return gemm.wgrad384(gy, x)
    if (USE_HIP_WGRAD and gy.is_cuda and gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and gy.is_contiguous()
            and x.is_contiguous() and n % 8 == 0 and k % 8 == 0 and t < (1 << 21)):
        return conv_gemm.linear_wgrad(x, gy)
    b, size = _chunks(t, n, k)
    main = b * size
    out = torch.bmm(gy[:main].view(b, size, n).transpose(1, 2), x[:main].view(b, size, k),
                    out_dtype=torch.float32).sum(0)
    if main < t:
        out = out + torch.mm(gy[main:].t(), x[main:], out_dtype=torch.float32)
    return out


def weight_bias_grad(gy, x, want_w, want_b):
    """(dW fp32 or None, db fp32 or None) of a linear layer over tokens; both from ONE pass over gy where the
    token-streaming kernel applies (gemm.wgrad384: the column sums come out of the fragments it holds anyway), else the
    weight gradient above and rows.colsum_any."""
    if want_w and want_b and USE_HIP_WGRAD and gemm.wgrad384_usable(gy, x):
        return gemm.wgrad384(gy, x, with_bias=True)
    t, n = gy.shape
    k = x.shape[1]
    if (want_w and want_b and USE_HIP_WGRAD and gy.is_cuda and gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
            and gy.is_contiguous() and x.is_contiguous() and n % 8 == 0 and k % 8 == 0 and t < (1 << 21)
            and not gemm.wgrad384_usable(gy, x) and conv_gemm.linear_wgrad_bias_usable(k, n)):
        return conv_gemm.linear_wgrad_bias(x, gy)       # the Swin widths: the column sums ride in the product's padding column
    return (weight_grad(gy, x) if want_w else None), (rows.colsum_any(gy) if want_b else None)


class _TokenLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, force_hip=False, weight2=None):
        """weight2: a second parameter stacked under `weight` (one GEMM for two projections of the same input); its
        rows follow weight's in the output, `bias` is then the caller's concatenation."""
        xb = x.to(torch.bfloat16)
        # the step's bf16 mirror of the parameter (transoar_amd/shadow.py) instead of a cast kernel per call
        if weight2 is None:
            wb = shadow.bf16_or_cast(weight)
        else:
            wb = shadow.bf16_stack((weight, weight2))
            if wb is None:
                wb = torch.cat((weight, weight2)).to(torch.bfloat16)
        ctx.save_for_backward(xb, wb)
        shadow.stamp(ctx)
        ctx.in_dtype, ctx.has_bias, ctx.force_hip = x.dtype, bias is not None, force_hip
        ctx.split = None if weight2 is None else weight.shape[0]
        global LAST_PATH
        x2 = xb.reshape(-1, xb.shape[-1])
        if _hip_gemm(x2, wb, force_hip):
            LAST_PATH = "hip-gemm"
            # the bias is added in fp32 before the single rounding to bf16 (F.linear rounds the bias to bf16 first)
            return gemm.linear_nt(x2, wb, bias).view(*xb.shape[:-1], wb.shape[0])
        LAST_PATH = "blas"
        with torch.autocast("cuda", enabled=False):
            return F.linear(xb, wb, None if bias is None else shadow.bf16_or_cast(bias))

    @staticmethod
    def backward(ctx, gy):
        shadow.check(ctx)
        xb, wb = ctx.saved_tensors
        gy = gy.to(torch.bfloat16)
        gy2 = gy.reshape(-1, gy.shape[-1])
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        gx = gw = gb = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                wt = wb.t().contiguous()                   # (K, N): dX = dY . W as an NT product
                gx = gemm.linear_nt(gy2, wt) if _hip_gemm(gy2, wt, ctx.force_hip) else torch.mm(gy2, wb)
                gx = gx.view(xb.shape).to(ctx.in_dtype)
            gw, gb = weight_bias_grad(gy2, xb.reshape(-1, xb.shape[-1]),
                                      ctx.needs_input_grad[1] or (ctx.split is not None and ctx.needs_input_grad[4]),
                                      ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.split is not None and gw is not None:
            return gx, gw[:ctx.split], gb, None, gw[ctx.split:]
        return gx, gw, gb, None, None


class _LinearReluDropout(torch.autograd.Function):
    """dropout(relu(x W^T + b)) with bias, ReLU and the seeded dropout mask in the epilogue of the K = 384 streaming GEMM
    (decoder_blocks.py:166: ``self.dropout2(self.activation(self.linear1(src)))``): the 480-MB hidden tensor is written
    once instead of written, re-read and re-written.  The backward needs only the output (y > 0 <=> kept and pre-activation
    > 0): gh = y > 0 ? gy * scale : 0 (tokens' relu_dropout_backward), then the data / weight / bias gradients as usual."""

    @staticmethod
    def forward(ctx, x, weight, bias, seed, keep_prob):
        xb = x.to(torch.bfloat16)
        wb = shadow.bf16_or_cast(weight)
        x2 = xb.reshape(-1, xb.shape[-1])
        y = gemm.linear_relu_dropout(x2, wb, bias, seed, keep_prob)
        ctx.save_for_backward(xb, wb, y)
        shadow.stamp(ctx)
        ctx.in_dtype, ctx.has_bias, ctx.scale = x.dtype, bias is not None, 1.0 / keep_prob
        return y.view(*xb.shape[:-1], wb.shape[0])

    @staticmethod
    def backward(ctx, gy):
        from . import tokens
        shadow.check(ctx)
        xb, wb, y = ctx.saved_tensors
        gy2 = gy.to(torch.bfloat16).reshape(-1, gy.shape[-1]).contiguous()
        gh = torch.empty_like(y)
        with torch.cuda.device(y.device):
            rc = tokens.lib.transoar_relu_dropout_backward(gy2.data_ptr(), y.data_ptr(), ctx.scale, gh.data_ptr(), y.numel(),
                                                           torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("transoar_relu_dropout_backward failed with code %d" % rc)
        gx = gw = gb = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                wt = wb.t().contiguous()
                gx = gemm.linear_nt(gh, wt).view(xb.shape).to(ctx.in_dtype)
            gw, gb = weight_bias_grad(gh, xb.reshape(-1, xb.shape[-1]), ctx.needs_input_grad[1],
                                      ctx.has_bias and ctx.needs_input_grad[2])
        return gx, gw, gb, None, None


def linear_relu_dropout_usable(x, weight):
    """The fused linear1 of the refinement block: bf16 autocast on the GPU, 384 input channels, dense tokens."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and weight.dtype == torch.float32
            and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16):
        return False
    x2 = x.reshape(-1, x.shape[-1])
    # exactly the products gemm.stream_kind hands to the K = 384 kernel (a hidden width of 384 is not one of them unless
    # TRANSOAR_GEMM_STREAM_SQUARE is set), on operands the C entry points accept (16-byte aligned)
    return (USE_HIP_GEMM is not False and x2.shape[1] == 384 and weight.shape[1] == 384 and x2.data_ptr() % 16 == 0
            and gemm.k384_takes(x2.shape[0], weight.shape[0]))


def linear_relu_dropout(x, weight, bias, dropout):
    """dropout(relu(linear(x))) in one GEMM launch; `dropout`: the nn.Dropout module (its p and training flag)."""
    from . import tokens
    active = dropout.training and dropout.p > 0.0
    seed = tokens.dropout_seed(x) if active else None
    return _LinearReluDropout.apply(x, weight, bias, seed, 1.0 - dropout.p if active else 1.0)


class _FusedFFN(torch.autograd.Function):
    """linear2(dropout(relu(linear1(x)))) as ONE autograd node (decoder_blocks.py:166-167): forward = the fused first layer
    above + the tiled GEMM; backward without a stand-alone element-wise pass --
        gh   = hidden > 0 ? (gy W2) / keep : 0      one K = 384 GEMM, gate in its epilogue (gemm.linear_gate)
        gW2, gb2 = gy^T hidden, sum gy               one token-streaming pass (gemm.wgrad384 with its column sums)
        gx   = gh W1                                 tiled GEMM
        gW1, gb1 = gh^T x, sum gh                    one token-streaming pass
    (round 4 before this: relu_dropout_backward read gy_hidden and hidden and wrote gh, 1.4 GB per layer, and two column-sum
    kernels re-read gy and gh)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, seed, keep_prob):
        xb = x.to(torch.bfloat16)
        w1b, w2b = shadow.bf16_or_cast(w1), shadow.bf16_or_cast(w2)
        x2 = xb.reshape(-1, xb.shape[-1])
        hidden = gemm.linear_relu_dropout(x2, w1b, b1, seed, keep_prob)
        out = gemm.linear_nt(hidden, w2b, b2)
        ctx.save_for_backward(xb, w1b, w2b, hidden)
        shadow.stamp(ctx)
        ctx.in_dtype, ctx.scale = x.dtype, 1.0 / keep_prob
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        return out.view(*xb.shape[:-1], w2b.shape[0])

    @staticmethod
    def backward(ctx, gy):
        shadow.check(ctx)
        xb, w1b, w2b, hidden = ctx.saved_tensors
        gy2 = gy.to(torch.bfloat16).reshape(-1, gy.shape[-1]).contiguous()
        x2 = xb.reshape(-1, xb.shape[-1])
        need = ctx.needs_input_grad
        gx = gw1 = gb1 = gw2 = gb2 = None
        with torch.autocast("cuda", enabled=False):
            gh = gemm.linear_gate(gy2, w2b.t().contiguous(), hidden, ctx.scale)
            gw2, gb2 = weight_bias_grad(gy2, hidden, need[3], ctx.has_b2 and need[4])
            if need[0]:
                gx = gemm.linear_nt(gh, w1b.t().contiguous()).view(xb.shape).to(ctx.in_dtype)
            gw1, gb1 = weight_bias_grad(gh, x2, need[1], ctx.has_b1 and need[2])
        return gx, gw1, gb1, gw2, gb2, None, None


FUSED_FFN = os.environ.get("TRANSOAR_FUSED_FFN", "1") != "0"


def fused_ffn_usable(x, w1, w2):
    """Both layers of the refinement block's FFN as one node: what the fused first layer needs, 384 channels in and out,
    a hidden width the K = 384 kernels take."""
    return (FUSED_FFN and linear_relu_dropout_usable(x, w1) and w2.dtype == torch.float32 and w2.shape[0] == 384
            and w2.shape[1] == w1.shape[0] and w1.shape[0] % 64 == 0)


def fused_ffn(x, linear1, linear2, dropout):
    """linear2(dropout(relu(linear1(x)))); `dropout`: the nn.Dropout module between the layers."""
    from . import tokens
    active = dropout.training and dropout.p > 0.0
    seed = tokens.dropout_seed(x) if active else None
    return _FusedFFN.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, seed, 1.0 - dropout.p if active else 1.0)


class _GeluMlp(torch.autograd.Function):
    """fc2(gelu(fc1(x))) of a Swin block (encoder_blocks.py Mlp) as ONE autograd node on the tiled GEMM, the activation in the
    epilogues on either side of it: forward = fc1 with (h, gelu(h)) out + fc2; backward
        gh = (gy W2) * gelu'(h)        one GEMM (gemm.linear_gelu_grad)        gW2, gb2 = gy^T a, sum gy
        gx = gh W1                                                              gW1, gb1 = gh^T x, sum gh
    -- torch's gelu / gelu_backward kernels (read h, write a; read ga and h, write gh: 3.1 GB per stage-0 block) are gone, and
    the arithmetic is theirs (fp32 on the bf16-rounded products)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        xb = x.to(torch.bfloat16)
        w1b, w2b = shadow.bf16_or_cast(w1), shadow.bf16_or_cast(w2)
        x2 = xb.reshape(-1, xb.shape[-1])
        h, a = gemm.linear_gelu(x2, w1b, b1)
        out = gemm.linear_nt(a, w2b, b2)
        ctx.save_for_backward(xb, w1b, w2b, h, a)
        shadow.stamp(ctx)
        ctx.in_dtype = x.dtype
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        return out.view(*xb.shape[:-1], w2b.shape[0])

    @staticmethod
    def backward(ctx, gy):
        shadow.check(ctx)
        xb, w1b, w2b, h, a = ctx.saved_tensors
        gy2 = gy.to(torch.bfloat16).reshape(-1, gy.shape[-1]).contiguous()
        x2 = xb.reshape(-1, xb.shape[-1])
        need = ctx.needs_input_grad
        gx = None
        with torch.autocast("cuda", enabled=False):
            gh = gemm.linear_gelu_grad(gy2, w2b.t().contiguous(), h)
            gw2, gb2 = weight_bias_grad(gy2, a, need[3], ctx.has_b2 and need[4])
            if need[0]:
                gx = gemm.linear_nt(gh, w1b.t().contiguous()).view(xb.shape).to(ctx.in_dtype)
            gw1, gb1 = weight_bias_grad(gh, x2, need[1], ctx.has_b1 and need[2])
        return gx, gw1, gb1, gw2, gb2


GELU_MLP = os.environ.get("TRANSOAR_GELU_MLP", "1") != "0"


def gelu_mlp_usable(x, fc1, fc2, min_tokens=None):
    """bf16 autocast on the GPU, fp32 parameters, enough tokens, shapes the tiled GEMM and its GELU epilogues take."""
    tokens = x.numel() // x.shape[-1]
    k, hid, n = fc1.weight.shape[1], fc1.weight.shape[0], fc2.weight.shape[0]
    return (GELU_MLP and x.is_cuda and tokens >= (MIN_TOKENS if min_tokens is None else min_tokens) and x.is_contiguous()
            and fc1.weight.dtype == torch.float32 and fc2.weight.dtype == torch.float32 and fc2.weight.shape[1] == hid
            and k % 8 == 0 and hid % 8 == 0 and n % 8 == 0 and tokens * max(hid, k, n) * 2 < 0x7ffffff0
            and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16)


def gelu_mlp(x, fc1, fc2):
    """fc2(gelu(fc1(x))) (exact GELU, no dropout between the layers)."""
    return _GeluMlp.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)


def token_linear(x, weight, bias=None, force_hip=False, min_tokens=None, weight2=None):
    """F.linear for (…, T, K) token tensors; the chunked-wgrad path applies to
    bf16 autocast on the GPU with enough tokens, the stock one otherwise.  force_hip: the hand-written GEMM for
    the forward and the data gradient whatever the shape (the FPN's 1x1x1 and transposed convolutions).
    weight2: a second weight parameter stacked under `weight` (y = x [weight; weight2]^T + bias, bias already stacked)."""
    tokens = x.numel() // x.shape[-1]
    if (x.is_cuda and tokens >= (MIN_TOKENS if min_tokens is None else min_tokens)
            and weight.dtype == torch.float32 and x.is_contiguous()
            and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16):
        return _TokenLinear.apply(x, weight, bias, force_hip, weight2)
    return F.linear(x, weight if weight2 is None else torch.cat((weight, weight2)), bias)
