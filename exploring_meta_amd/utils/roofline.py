"""Algorithmic work of one engine launch (SURVEY.md 8d accounting), for any ModelSpec: what bench.py's `roofline` object and
tools/roofline_table.py divide by the measured launch duration.

FLOPs of a conv-shaped op = 2*9*Ci*Co per output pixel per GEMM term; bytes = the tensors the op must read or write once (layer
input, conv output z, pooled output p; parameters are negligible).  Block 1 of a pooling net with 1 or 3 input channels runs
fused ("layer" 0): its forward / tangent-forward kernels recompute the convolution (FLOPs) and touch only the input, the pooled
output, zhat at the argmax and the argmax byte (bytes); its BatchNorm-backward reductions stream pooled tensors; its weight
gradients are the sparse matrix pass over pooling windows (algorithmic weight-gradient FLOPs)."""

PEAK_TFLOPS = 157.3      # fp32-input MFMA == fp32 vector peak (MI355X_MICROARCH.md)
PEAK_GBPS = 8000.0       # HBM3E spec (about 6300 GB/s is achievable with a streaming copy)
BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)
# The 32-channel stride-1 forward / dgrad convolutions run in the split-bf16 operand form (csrc/conv_mfma.hip): SIX bf16 products per
# algorithmic fp32 multiply-add, so the matrix pipe bounds them at a sixth of its dense bf16 rate, in algorithmic (fp32) FLOPs:
SPLIT_BF16_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 6.0
# ... or, when the OPT-IN two-plane fp16 form is selected (mi_conv_set_split_bf16(2), csrc/bf16_split.h; never the default): THREE fp16
# products per multiply-add (the fp16 MFMAs run at the bf16 rate), a third of the dense rate:
SPLIT_F16_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 3.0
SPLIT_BF16_OPS = ('conv_fwd_stats', 'dgrad', 'tangent_conv_fwd', 'tangent_dgrad')
SPLIT_BF16_WGRAD_OPS = ('wgrad', 'tangent_wgrad')      # csrc/wgrad_bf16.hip: maps at least 16 wide (the 10 x 10 block keeps the fp32 kernel)


def mfma_peak(spec, op, layer, form=1):
    """(peak TFLOP/s in algorithmic fp32 FLOPs, pipe) of the matrix pipe `op` on block `layer` runs on.  form: the operand form of the
    stride-1 hidden blocks (mi_conv_get_split_bf16): 0 / False fp32 pipe, 1 / True three bf16 planes, 2 two scaled fp16 planes."""
    h, w, ci, co, ho, wo, _, _ = layer_geometry(spec)[layer]
    same = ci == co and (h, w) == (ho, wo)                  # stride-1 hidden -> hidden block
    form = int(form)
    split = (SPLIT_F16_PEAK_TFLOPS, 'fp16 x3 (two scaled planes)') if form == 2 else (SPLIT_BF16_PEAK_TFLOPS, 'bf16 x6 (split operands)')
    if form and same and ci == 32 and (op in SPLIT_BF16_OPS or (op in SPLIT_BF16_WGRAD_OPS and w >= 16)):
        return split
    if form and same and ci == 64 and (op in ('conv_fwd_stats', 'dgrad') or (op in SPLIT_BF16_WGRAD_OPS and w >= 16)):
        # one-term forward / dgrad (two terms would need 216 / 144 KB of weight planes); weight gradients as at 32 filters
        return split
    if form and layer == 0 and ci == 3 and op in ('bn_relu_pool_fwd', 'bn_tangent_fwd'):
        # block 1's lean forward kernels (csrc/block1.hip): conv1 recomputed with EIGHT bf16 products per multiply-add (raw-pixel inputs: the two
        # 2^-24 cross terms stay in); the tangent forward since round 4, the forward since the end of round 6
        return BF16_PEAK_TFLOPS / 8.0, 'bf16 x8 (split operands)'
    if form and layer == 0 and op in SPLIT_BF16_WGRAD_OPS and ci == 3 and w == 84:
        # block 1's sparse weight gradient on 84-wide RGB inputs (csrc/gram.hip sparse_wgrad_rows_kernel<..., BF>): six bf16 products per
        # multiply-add laid out along K, whichever split form the hidden blocks use (round 6)
        return SPLIT_BF16_PEAK_TFLOPS, 'bf16 x6 (split operands)'
    return PEAK_TFLOPS, 'fp32'


def layer_geometry(spec):
    """[(h, w, ci, co, ho, wo, hp, wp)] per ConvBlock."""
    out, h, w, ci = [], spec.in_h, spec.in_w, spec.in_channels
    for _ in range(spec.n_layers):
        s = 1 if spec.max_pool else 2
        ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
        hp, wp = (ho // 2, wo // 2) if spec.max_pool else (ho, wo)
        out.append((h, w, ci, spec.hidden, ho, wo, hp, wp))
        h, w, ci = hp, wp, spec.hidden
    return out


def fused_block1(spec):
    return spec.max_pool and spec.in_channels in (1, 3) and spec.in_h % 2 == 0 and spec.in_w % 2 == 0 and spec.in_w // 2 >= 8


def op_costs(spec, op, layer, images):
    """(flops, bytes) of one launch of `op` on block `layer` (0-based) over `images` images, or None for latency-class ops."""
    h, w, ci, co, ho, wo, hp, wp = layer_geometry(spec)[layer]
    x, z, p = h * w * ci * 4, ho * wo * co * 4, hp * wp * co * 4
    f = 2 * 9 * ci * co * ho * wo
    if layer == 0 and fused_block1(spec):
        a = p // 4
        t = {'conv_fwd_stats': (f, x), 'bn_relu_pool_fwd': (f, x + 2 * p + a), 'bn_bwd_reduce': (0, 3 * p), 'wgrad': (f, x + p + a),
             'tangent_conv_fwd': (2 * f, x), 'bn_tangent_fwd': (f, x + 3 * p + a), 'bn_tangent_bwd_reduce': (0, 5 * p),
             'tangent_wgrad': (f, x + 2 * p + a)}        # (x has no tangent: R{dW1} = x^T R{dz} is one product)
    else:
        # The dgrad of a stride-1 hidden block also forms the BatchNorm-backward sums of the block below in its epilogue (EPI_BRED):
        # it reads that block's p and zhat-at-argmax (tangent: + zhat-dot and the primal cotangent) at its output positions -- tensors
        # of the shape of this block's input x -- and the forward kernels of a block whose sums will ride above it also store zhat at
        # the argmax (one more pooled-shape tensor).
        rides = spec.max_pool and layer >= 1                 # this block's dgrad carries the sums of block layer-1
        ridden = spec.max_pool and layer + 1 < spec.n_layers   # this block's own sums ride in block layer+1's dgrad
        # (where the block below is the fused block 1, its argmax BYTE stands in for p: "ReLU on" <=> the byte is below 4)
        pb = x // 4 if (layer == 1 and fused_block1(spec)) else x
        t = {'conv_fwd_stats': (f, x + z), 'dgrad': (f, z + x + (x + pb if rides else 0)), 'wgrad': (f, x + z),
             'tangent_conv_fwd': (2 * f, 2 * x + 2 * z), 'tangent_dgrad': (2 * f, 2 * z + x + (3 * x + pb if rides else 0)),
             'tangent_wgrad': (2 * f, 2 * x + 2 * z),
             'bn_relu_pool_fwd': (0, z + p + (p if ridden else 0)), 'bn_bwd_reduce': (0, z + p), 'bn_bwd_apply': (0, 2 * z + p),
             'bn_tangent_fwd': (0, 2 * z + p + (p if ridden else 0)), 'bn_tangent_bwd_reduce': (0, 2 * z + 2 * p),
             'bn_tangent_bwd_apply': (0, 3 * z + 2 * p)}
        if layer == 0:          # first block without a tangent input: one GEMM term
            t['tangent_conv_fwd'] = (f, x + 2 * z)
            t['tangent_wgrad'] = (f, x + z)
    if op not in t:
        return None
    fl, by = t[op]
    return fl * images, by * images


def bound_of(flops, nbytes, peak_tflops=PEAK_TFLOPS):
    """Which roofline bounds an op with this arithmetic intensity (peak_tflops: the matrix pipe it runs on, mfma_peak)."""
    if not flops:
        return 'hbm'
    return 'mfma' if flops / nbytes > peak_tflops * 1e12 / (PEAK_GBPS * 1e9) else 'hbm'


KERNEL_NAMES = {
    'conv_fwd_stats': 'conv3x3_s1_mfma_kernel<{ci},1,EPI_STATS,fwd>', 'dgrad': 'conv3x3_s1_mfma_kernel<{ci},1,EPI_BRED,dgrad>',
    'wgrad': 'wgrad3x3_rows_mfma_kernel (1 term)', 'tangent_conv_fwd': 'conv3x3_s1_mfma_kernel<{ci},2,EPI_TSTATS,fwd>',
    'tangent_dgrad': 'conv3x3_s1_mfma_kernel<{ci},2,EPI_BRED,dgrad>', 'tangent_wgrad': 'wgrad3x3_rows_mfma_kernel (2 terms)',
    'bn_relu_pool_fwd': 'bn_fwd_kernel', 'bn_bwd_reduce': 'bn_bwd_reduce_kernel', 'bn_bwd_apply': 'bn_bwd_apply_kernel',
    'bn_tangent_fwd': 'bn_tan_fwd_kernel', 'bn_tangent_bwd_reduce': 'bn_tan_bwd_reduce_kernel',
    'bn_tangent_bwd_apply': 'bn_tan_bwd_apply_kernel',
}
BLOCK1_KERNEL_NAMES = {
    'bn_relu_pool_fwd': 'block1_fwd_kernel<{ci},FWD>', 'bn_tangent_fwd': 'block1_fwd_kernel<{ci},TFWD_ARG>', 'wgrad': 'sparse_wgrad_rows_kernel<{ci},false>',
    'tangent_wgrad': 'sparse_wgrad_rows_kernel<{ci},true>', 'bn_bwd_reduce': 'pooled_reduce_kernel<false>',
    'bn_tangent_bwd_reduce': 'pooled_reduce_kernel<true>', 'conv_fwd_stats': 'block1_kernel<{ci},STATS>',
}


def conv_kernel_is_b16(mpix, tasks, co, mode, min_tpw=6):
    """Whether a split-bf16 stride-1 convolution launch over `tasks` tasks of `mpix` output pixels each takes the 16x16x32 kernel
    (csrc/conv_b16.h) -- launch_conv3x3's rule: mi_conv_set_b16 mode 2 always, mode 1 from `min_tpw` tiles of 30 pixels per wave
    (2048 resident waves) on, mode 0 never."""
    if mode == 2:
        return True
    if mode != 1:
        return False
    tiles = -(-mpix // 30) * tasks * max(1, co // 32)
    return -(-tiles // 2048) >= min_tpw


def kernel_name(spec, op, layer, b16=False):
    ci = layer_geometry(spec)[layer][2]
    names = BLOCK1_KERNEL_NAMES if (layer == 0 and fused_block1(spec)) else KERNEL_NAMES
    name = names.get(op, op).format(ci=ci)
    if b16 and name.startswith('conv3x3_s1_mfma_kernel'):
        name = name.replace('conv3x3_s1_mfma_kernel', 'conv3x3_s1_b16_kernel')
    return name
