"""Op-level parity of the HIP kernels (through the C-ABI) against a plain torch fp32 reference (conv / pool /
pack) and against the oracle (NMS, RPN selection, ROIAlign, box-head post-processing)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import proben_amd  # noqa: F401
    from proben_amd import layers
    return layers


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # (N, H, W, Cin, Cout, kernel, stride, relu, res_mode, f32out)
    (2, 25, 32, 256, 256, 3, 1, True, 0, False),
    (1, 50, 64, 64, 64, 3, 1, True, 0, False),
    (2, 13, 16, 512, 128, 3, 1, False, 0, False),
    (2, 50, 64, 256, 1024, 1, 1, True, 1, False),
    (2, 50, 64, 1024, 256, 1, 1, True, 0, False),
    (3, 51, 63, 256, 512, 1, 2, False, 0, False),
    (2, 26, 32, 512, 256, 1, 1, False, 2, False),
    (2, 20, 24, 256, 15, 1, 1, False, 0, True),
    (1, 37, 41, 64, 256, 1, 1, False, 0, False),
    (16, 100, 128, 128, 128, 3, 1, True, 0, False),   # >= 512 tiles of 256 rows: the 8-wave 256x128 3x3 kernel
    (9, 99, 131, 64, 256, 3, 1, False, 0, False),     # same kernel, ragged M / odd width (edge masks, M tail)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_torch(L, case):
    N, H, W, Cin, Cout, k, s, relu, res_mode, f32out = case
    g = torch.Generator(device="cpu").manual_seed(hash(case) % 2**31)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, stride=s, padding=k // 2)
    res = None
    if res_mode == 1:
        res = torch.randn(ref.shape, generator=g).cuda().half()
        ref = ref + res.float()
        res = nhwc(res)
    elif res_mode == 2:
        res = torch.randn(N, Cout, (ref.shape[2] + 1) // 2, (ref.shape[3] + 1) // 2, generator=g).cuda().half()
        ref = ref + torch.nn.functional.interpolate(res.float(), scale_factor=2, mode="nearest")[:, :, : ref.shape[2], : ref.shape[3]]
        res = nhwc(res)
    if relu:
        ref = ref.relu()
    wp = w.permute(0, 2, 3, 1).contiguous()
    if f32out:
        out = L.conv2d_nhwc(nhwc(x), wp, b, kernel=k, stride=s, relu=relu, out_f32=True, cout=Cout, cout_store=Cout, out_stride=16)
        got = out[..., :Cout].permute(0, 3, 1, 2)
        tol = dict(rtol=1e-3, atol=1e-3)
    else:
        out = L.conv2d_nhwc(nhwc(x), wp, b, kernel=k, stride=s, relu=relu, residual=res, residual_mode=res_mode)
        got = out.permute(0, 3, 1, 2).float()
        tol = dict(rtol=4e-3, atol=4e-3)
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, **tol)


@pytest.mark.parametrize("case", [(8, 100, 128, 128, 256, 3, 1, True, 0), (9, 99, 131, 256, 512, 1, 1, True, 1),
                                  (8, 100, 128, 256, 256, 1, 1, False, 2), (16, 51, 64, 512, 256, 1, 2, False, 0),
                                  (8, 100, 128, 64, 256, 1, 1, True, 0), (8, 100, 128, 128, 256, 1, 1, False, 1)])
@pytest.mark.parametrize("policy", [25])
def test_conv_big_tile_kernel(L, case, policy):
    """The 256x256 two-stage kernel (tile policy bit 3) against torch, incl. ragged M, residual modes, stride 2."""
    import proben_amd
    N, H, W, Cin, Cout, k, s, relu, res_mode = case
    hooks = proben_amd._lib.test_hooks()
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, stride=s, padding=k // 2)
    res = None
    if res_mode == 1:
        res = torch.randn(ref.shape, generator=g).cuda().half()
        ref = ref + res.float()
        res = nhwc(res)
    elif res_mode == 2:
        res = torch.randn(N, Cout, (ref.shape[2] + 1) // 2, (ref.shape[3] + 1) // 2, generator=g).cuda().half()
        ref = ref + torch.nn.functional.interpolate(res.float(), scale_factor=2, mode="nearest")[:, :, : ref.shape[2], : ref.shape[3]]
        res = nhwc(res)
    if relu:
        ref = ref.relu()
    # 25: 256x256 two-stage kernel everywhere it applies
    hooks.pe_test_set_conv_policy(policy, 1)
    try:
        out = L.conv2d_nhwc(nhwc(x), w.permute(0, 2, 3, 1).contiguous(), b, kernel=k, stride=s, relu=relu, residual=res, residual_mode=res_mode)
        torch.cuda.synchronize()
    finally:
        hooks.pe_test_set_conv_policy(L.DEFAULT_CONV_POLICY, 1)
    torch.testing.assert_close(out.permute(0, 3, 1, 2).float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, relu[, stride]
    (3, 50, 64, 1024, 256, True),       # res4 conv1: 300 blocks of 32 pixels over 256 workgroups (runs of 1 and 2 blocks)
    (32, 50, 64, 1024, 256, True),      # ... at the bench's batch: runs of 12 / 13 blocks = 3 full tiles + a 1-block tile
    (1, 13, 16, 2048, 256, False),      # M = 208 is not a multiple of 32: the last block is ragged, 7 workgroups
    (2, 25, 32, 2048, 512, True),       # two column tiles walked by ONE workgroup per run (grid too small to pair them)
    (4000, 1, 1, 1024, 1024, True),     # four column tiles on four workgroups of one XCD (fc2 shape)
    (1, 7, 9, 512, 256, True),          # two blocks
    (8, 100, 128, 256, 512, False, 2),  # stride 2: res3's shortcut convolution (K = 256), two column tiles
    (3, 51, 65, 512, 1024, True, 2),    # stride 2 on odd sizes: 26 x 33 outputs, rows of a tile straddle image rows and images
    (2, 50, 64, 1024, 2048, False, 2),  # res5's shortcut: eight column tiles
    (8, 100, 128, 128, 512, True, 1, 1),    # res3 conv3 + identity (residual mode 1), K = 128: two K-steps per tile
    (3, 51, 65, 256, 256, False, 1, 2),     # FPN lateral + nearest-2x top-down map (mode 2) on odd sizes: 26 x 33 map
    (4, 25, 32, 512, 2048, True, 1, 1),     # res5 conv3 + identity, eight column tiles
    (16, 100, 128, 512, 256, True, 1, 2),   # p3 lateral at a size where tiles straddle image rows
])
def test_conv1x1_ring_kernel_matches_torch_and_conv_igemm2_bit_for_bit(L, case):
    """csrc/conv1x1_ring.hip (persistent loader / consumer 1x1 kernel, the default for 1x1 layers with a bias and Cout % 256 == 0: residual-free
    from K = 512 at stride 1 / K = 256 at stride 2, with a residual - both modes - from K = 128) against torch fp32 - and bit for bit against conv_igemm2 / conv_big, the kernels
    the same layers ran on before and still run on without a bias: the choice between them must never show in a frame's result.  Any
    number of workgroups walks the same pixels to the same bits."""
    import proben_amd
    N, H, W, Cin, Cout, relu = case[:6]
    stride = case[6] if len(case) > 6 else 1
    res_mode = case[7] if len(case) > 7 else 0
    hooks = proben_amd._lib.test_hooks()
    g = torch.Generator(device="cpu").manual_seed(21)
    x = torch.randn(N, H, W, Cin, generator=g).cuda().half().relu()
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    Mo = N * Ho * Wo
    assert L.conv_variant_name(Mo, Cout, 1, Cin, stride=stride, residual_mode=res_mode, in_pixels=N * H * W) == "conv1x1_ring_kernel"
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).float(), w.permute(0, 3, 1, 2).float(), b, stride=stride)
    res = None
    if res_mode == 1:
        res = torch.randn(N, Ho, Wo, Cout, generator=g).cuda().half()
        ref = ref + res.permute(0, 3, 1, 2).float()
    elif res_mode == 2:
        res = torch.randn(N, (Ho + 1) // 2, (Wo + 1) // 2, Cout, generator=g).cuda().half()
        ref = ref + torch.nn.functional.interpolate(res.permute(0, 3, 1, 2).float(), scale_factor=2, mode="nearest")[:, :, :Ho, :Wo]
    ref = (ref.relu() if relu else ref).permute(0, 2, 3, 1)
    try:
        hooks.pe_test_set_conv_policy(9, 1)                      # round-4 dispatch: conv_igemm2 / conv_big
        old = L.conv2d_nhwc(x, w, b, kernel=1, stride=stride, relu=relu, residual=res, residual_mode=res_mode)
        hooks.pe_test_set_conv_policy(L.DEFAULT_CONV_POLICY, 1)
        out = torch.full_like(old, float("nan"))
        L.conv2d_nhwc(x, w, b, kernel=1, stride=stride, relu=relu, residual=res, residual_mode=res_mode, out=out)
        outs = [out.clone()]
        for wgs in (8, 64, 248):
            hooks.pe_test_set_ring_wgs(wgs)
            o = torch.full_like(old, float("nan"))
            L.conv2d_nhwc(x, w, b, kernel=1, stride=stride, relu=relu, residual=res, residual_mode=res_mode, out=o)
            outs.append(o)
        torch.cuda.synchronize()
    finally:
        hooks.pe_test_set_ring_wgs(256)
        hooks.pe_test_set_conv_policy(L.DEFAULT_CONV_POLICY, 1)
    torch.testing.assert_close(out.float(), ref, rtol=4e-3, atol=4e-3)
    for o in outs:
        assert torch.equal(o, old)


@pytest.mark.parametrize("case", [(16, 100, 128, 128, 128, 3, 1, True, 0), (9, 99, 131, 64, 256, 3, 1, False, 0),
                                  (2, 25, 32, 256, 256, 3, 1, True, 0), (3, 40, 50, 64, 64, 3, 1, True, 0)])
def test_conv3x3_weight_double_buffered_kernel(L, case):
    """The kw-reuse 3x3 kernel with the double-buffered weight tile, 128- and 256-row tiles."""
    import proben_amd
    N, H, W, Cin, Cout, k, s, relu, rows = case
    hooks = proben_amd._lib.test_hooks()
    g = torch.Generator(device="cpu").manual_seed(13)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, stride=s, padding=1)
    if relu:
        ref = ref.relu()
    hooks.pe_test_set_conv_policy(9, 1)
    try:
        out = L.conv2d_nhwc(nhwc(x), w.permute(0, 2, 3, 1).contiguous(), b, kernel=k, stride=s, relu=relu)
        torch.cuda.synchronize()
    finally:
        hooks.pe_test_set_conv_policy(L.DEFAULT_CONV_POLICY, 1)
    torch.testing.assert_close(out.permute(0, 3, 1, 2).float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, relu, out window (offset, stride)
    (3, 50, 64, 256, 256, True, None),      # res4 geometry: two image rows per tile, tiles straddle images
    (2, 20, 256, 64, 256, False, None),     # p2 geometry: half a row per tile (left / right halo from the neighbour tile)
    (3, 5, 32, 128, 512, True, None),       # four rows per tile, M = 480 is not a multiple of the 128-pixel tile, two N tiles
    (2, 13, 128, 256, 256, False, (256, 512)),  # one row per tile; output written into a channel window (middle fusion)
    (1, 7, 384, 64, 256, True, None),       # W = 3 x 128
    (4, 25, 32, 512, 512, True, None),      # res5 geometry
])
def test_conv3x3_weights_direct_kernel(L, case):
    """csrc/conv_wd.h (weights streamed L2 -> VGPR in fragment order, pixel slab with explicit halo in LDS) against
    torch fp32: image borders, tile seams inside and across image rows, ragged M, bias folded into the accumulators,
    channel-window output.  Run twice (race screen: the LDS ring is published by one barrier per 12 K-steps)."""
    N, H, W, Cin, Cout, relu, window = case
    g = torch.Generator(device="cpu").manual_seed(23)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, stride=1, padding=1)
    if relu:
        ref = ref.relu()
    assert L.conv_wd_supported(3, 1, H, W, Cin, Cout)
    packed = L.conv_wd_pack(w.permute(0, 2, 3, 1).contiguous())
    outs = []
    for _ in range(2):
        if window is None:
            o = L.conv3x3_wd(nhwc(x), packed, b, Cout, relu=relu)
        else:
            off, stride = window
            full = torch.full((N, H, W, stride), 7.0, dtype=torch.float16, device="cuda")
            L.conv3x3_wd(nhwc(x), packed, b, Cout, relu=relu, out=full.view(-1)[off:], out_stride=stride)
            assert torch.all(full[..., :off] == 7.0)          # the other half of the window is untouched
            o = full[..., off:off + Cout]
        outs.append(o.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    torch.testing.assert_close(outs[0].permute(0, 3, 1, 2).float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, relu, out window (offset, stride)
    (3, 50, 64, 256, 256, True, None),          # res4 / p4 geometry: four image rows per tile, tiles straddle images, ragged last tile
    (2, 20, 256, 64, 256, False, None),         # p2 geometry: one row per tile, three K groups only (the slab ring wraps inside a tile)
    (2, 13, 128, 256, 256, False, (256, 512)),  # p3 geometry, odd row count (last tile has one row), channel-window output
    (1, 9, 64, 128, 512, True, None),           # two channel tiles: the bias / weight stream switch between a workgroup's tiles
    (40, 7, 64, 64, 256, True, None),           # more tiles than one round of workgroups would hold in the 8 per-XCD runs, H < 8
])
def test_conv3x3_wd9_is_bit_identical_to_conv_wd(L, case):
    """csrc/conv_wd9.h (persistent workgroups, one wave per SIMD, accumulators in a[0:255] from inline-asm MFMAs, the slab by
    LDS-DMA with an XOR swizzle, whole-line stores) against csrc/conv_wd.h ON THE SAME CALL: bit for bit, twice (race screen for
    the counted vmcnt / barrier protocol of the DMA ring), and against torch fp32."""
    from proben_amd import _lib
    N, H, W, Cin, Cout, relu, window = case
    g = torch.Generator(device="cpu").manual_seed(29)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda().half()
    b = torch.randn(Cout, generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, stride=1, padding=1)
    if relu:
        ref = ref.relu()
    packed = L.conv_wd_pack(w.permute(0, 2, 3, 1).contiguous())
    hooks = _lib.test_hooks()

    def run():
        if window is None:
            return L.conv3x3_wd(nhwc(x), packed, b, Cout, relu=relu).clone()
        off, stride = window
        full = torch.full((N, H, W, stride), 7.0, dtype=torch.float16, device="cuda")
        L.conv3x3_wd(nhwc(x), packed, b, Cout, relu=relu, out=full.view(-1)[off:], out_stride=stride)
        assert torch.all(full[..., :off] == 7.0)
        return full[..., off:off + Cout].clone()
    try:
        hooks.pe_test_set_wd9_mode(0)
        old = run()
        hooks.pe_test_set_wd9_mode(2)
        new1, new2 = run(), run()
    finally:
        hooks.pe_test_set_wd9_mode(-1)
    torch.cuda.synchronize()
    assert torch.equal(new1, new2)
    assert torch.equal(new1, old)
    torch.testing.assert_close(new1.permute(0, 3, 1, 2).float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("shape", [(2, 40, 64, 256), (1, 13, 128, 256), (2, 9, 256, 256), (1, 5, 256, 64), (3, 3, 64, 128), (33, 2, 128, 256)])
def test_fused_rpn_head_wd9_is_bit_identical_to_conv_wd(L, shape):
    """The fused RPN head on the one-wave structure (csrc/conv_wd9.h, EP = 2: t stays in the accumulators, which ARE the head MFMA's
    B fragments; cross-wave sum through LDS in the two-wave kernel's order) against conv_wd.h's HEAD == 1: the same bits, for every
    width, ragged last tiles (H not a multiple of the tile's rows), image seams inside tiles, more tiles than workgroups, twice."""
    from proben_amd import _lib
    N, H, W, Cin = shape
    g = torch.Generator(device="cpu").manual_seed(43)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w3 = (torch.randn(256, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda().half()
    b3 = torch.randn(256, generator=g).cuda()
    wh = (torch.randn(15, 256, 1, 1, generator=g) / 16.0).cuda().half()
    b16 = torch.zeros(16, device="cuda")
    b16[:15] = torch.randn(15, generator=g).cuda()
    pk = L.conv_wd_pack(w3.permute(0, 2, 3, 1).contiguous())
    ph = L.conv_wd_pack_head(wh.reshape(15, 256).contiguous())
    hooks = _lib.test_hooks()
    run = lambda: L.conv3x3_wd_rpn_head(nhwc(x), pk, b3, ph, b16).clone()
    try:
        hooks.pe_test_set_wd9_mode(0)
        old = run()
        hooks.pe_test_set_wd9_mode(2 | 8)
        new1, new2 = run(), run()
        hooks.pe_test_set_wd9_wgs(8, 0)          # 8 workgroups: every workgroup walks many tiles (the persistent loop's ring hand-over)
        few = run()
    finally:
        hooks.pe_test_set_wd9_wgs(256, 0)
        hooks.pe_test_set_wd9_mode(-1)
    torch.cuda.synchronize()
    assert torch.equal(new1, new2) and torch.equal(new1, few)
    assert torch.equal(new1, old)
    t_ref = torch.nn.functional.conv2d(x.float(), w3.float(), b3, padding=1).relu()
    ref = torch.nn.functional.conv2d(t_ref, wh.float(), b16[:15]).permute(0, 2, 3, 1)
    torch.testing.assert_close(new1[..., :15], ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("shape", [(2, 40, 64, 256), (1, 13, 128, 256), (3, 5, 32, 256), (2, 8, 256, 512)])
def test_fused_rpn_head_matches_two_launch_path_and_torch(L, shape):
    """StandardRPNHead (proposal_generator/rpn.py:74-85) in one launch == 3x3 + ReLU then the 15-column 1x1 (the fp16
    rounding of t is the same; only the fp32 summation order over the four 64-channel slices differs) == torch fp32."""
    N, H, W, Cin = shape
    g = torch.Generator(device="cpu").manual_seed(31)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half()
    w3 = (torch.randn(256, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda().half()
    b3 = torch.randn(256, generator=g).cuda()
    wh = (torch.randn(15, 256, 1, 1, generator=g) / 16.0).cuda().half()
    bh = torch.randn(15, generator=g).cuda()
    t_ref = torch.nn.functional.conv2d(x.float(), w3.float(), b3, padding=1).relu()
    ref = torch.nn.functional.conv2d(t_ref, wh.float(), bh).permute(0, 2, 3, 1)
    pk = L.conv_wd_pack(w3.permute(0, 2, 3, 1).contiguous())
    ph = L.conv_wd_pack_head(wh.reshape(15, 256).contiguous())
    b16 = torch.zeros(16, device="cuda")
    b16[:15] = bh
    outs = [L.conv3x3_wd_rpn_head(nhwc(x), pk, b3, ph, b16) for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    fused = outs[0]
    assert fused.shape == (N, H, W, 16) and float(fused[..., 15].abs().max()) == 0.0
    t = L.conv3x3_wd(nhwc(x), pk, b3, 256, relu=True)
    two = L.conv2d_nhwc(t, wh.permute(0, 2, 3, 1).contiguous(), bh, kernel=1, out_f32=True, cout=15, cout_store=15, out_stride=16)
    torch.testing.assert_close(fused[..., :15], two[..., :15], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(fused[..., :15], ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("shape", [(2, 50, 64, 256, 1024, True), (3, 5, 32, 256, 1024, True), (1, 9, 128, 128, 512, False), (2, 6, 256, 256, 256, True)])
def test_fused_bottleneck_tail_matches_two_launch_path_and_torch(L, shape):
    """BottleneckBlock's second half (backbone/resnet.py:207-221) in one launch == conv2 (weights-direct 3x3) + conv3 (LDS-DMA
    1x1 with residual) as two launches == torch fp32; ragged M (480 pixels), no-residual form, run twice (race screen: the
    fp16 copy of t reuses the slab ring after the loop's last barrier)."""
    N, H, W, Cin, CoutT, with_res = shape
    g = torch.Generator(device="cpu").manual_seed(37)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().half().relu()
    w2 = (torch.randn(256, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda().half()
    b2 = torch.randn(256, generator=g).cuda()
    w3 = (torch.randn(CoutT, 256, 1, 1, generator=g) / 16.0).cuda().half()
    b3 = torch.randn(CoutT, generator=g).cuda()
    res = torch.randn(N, CoutT, H, W, generator=g).cuda().half() if with_res else None
    t_ref = torch.nn.functional.conv2d(x.float(), w2.float(), b2, padding=1).relu()
    ref = torch.nn.functional.conv2d(t_ref, w3.float(), b3)
    if with_res:
        ref = ref + res.float()
    ref = ref.relu().permute(0, 2, 3, 1)
    p2 = L.conv_wd_pack(w2.permute(0, 2, 3, 1).contiguous())
    p3 = L.conv_wd_pack_tail(w3.reshape(CoutT, 256).contiguous())
    r = nhwc(res) if with_res else None
    outs = [L.bottleneck_tail_wd(nhwc(x), p2, b2, p3, b3, r, CoutT) for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    t = L.conv3x3_wd(nhwc(x), p2, b2, 256, relu=True)
    two = L.conv2d_nhwc(t, w3.permute(0, 2, 3, 1).contiguous(), b3, kernel=1, relu=True, residual=r, residual_mode=1 if with_res else 0)
    torch.testing.assert_close(outs[0].float(), two.float(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(outs[0].float(), ref, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("shape", [(2, 8, 64, True, True), (1, 7, 70, False, True), (3, 5, 32, False, False), (2, 13, 130, True, False), (1, 50, 64, False, True)])
def test_fused_bneck64_chain_matches_torch_and_unfused_kernels(L, shape):
    """A res2 bottleneck from its 3x3 on + the next block's conv1 (backbone/resnet.py:107-221) in one launch == torch fp32 with
    the fp16 roundings of the unfused pipeline (t2 and the block output are fp16 tensors there) == the unfused kernels;
    ragged tiles (7 x 70, 13 x 130 pixels vs 4 x 64 tiles), identity and convolution shortcuts, with / without the next conv1,
    run twice (race screen)."""
    N, H, W, sc, nxt = shape
    F = torch.nn.functional
    g = torch.Generator(device="cpu").manual_seed(41)
    t1 = torch.randn(N, 64, H, W, generator=g).cuda().half().relu()
    src = (torch.randn(N, 64 if sc else 256, H, W, generator=g).cuda().half()).relu()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).cuda().half()
    b2 = torch.randn(64, generator=g).cuda() * 0.5
    w3 = (torch.randn(256, 64, 1, 1, generator=g) / 8.0).cuda().half()
    b3 = torch.randn(256, generator=g).cuda() * 0.5
    wsc = (torch.randn(256, 64, 1, 1, generator=g) / 8.0).cuda().half() if sc else None
    bsc = torch.randn(256, generator=g).cuda() * 0.5 if sc else None
    w1n = (torch.randn(64, 256, 1, 1, generator=g) / 16.0).cuda().half() if nxt else None
    b1n = torch.randn(64, generator=g).cuda() * 0.5 if nxt else None
    t2 = F.conv2d(t1.float(), w2.float(), b2, padding=1).relu().half().float()
    short = F.conv2d(src.float(), wsc.float(), bsc) if sc else src.float()
    out_ref = (F.conv2d(t2, w3.float(), b3) + short).relu()
    ohwi = lambda w: w.permute(0, 2, 3, 1).contiguous()
    packed = L.bneck64_pack(ohwi(w2), ohwi(w3).reshape(256, 64), ohwi(wsc).reshape(256, 64) if sc else None, ohwi(w1n).reshape(64, 256) if nxt else None)
    runs = [L.bneck64(nhwc(t1), nhwc(src), packed, b2, b3, bsc, b1n) for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(runs[0][0], runs[1][0])
    out, t1n = runs[0]
    torch.testing.assert_close(out.float(), out_ref.permute(0, 2, 3, 1), rtol=5e-3, atol=5e-3)
    if nxt:
        assert torch.equal(runs[0][1], runs[1][1])
        t1n_ref = F.conv2d(out.permute(0, 3, 1, 2).float(), w1n.float(), b1n).relu()      # from the kernel's own fp16 block output
        torch.testing.assert_close(t1n.float(), t1n_ref.permute(0, 2, 3, 1), rtol=5e-3, atol=5e-3)
    else:
        assert t1n is None
    # the unfused kernels on the same data
    u2 = L.conv2d_nhwc(nhwc(t1), ohwi(w2), b2, kernel=3, relu=True)
    r = L.conv2d_nhwc(nhwc(src), ohwi(wsc), bsc, kernel=1) if sc else nhwc(src)
    u3 = L.conv2d_nhwc(u2, ohwi(w3), b3, kernel=1, relu=True, residual=r, residual_mode=1)
    torch.testing.assert_close(out.float(), u3.float(), rtol=4e-3, atol=4e-3)


def test_conv3x3_weights_direct_matches_lds_kernel_and_rejects_other_geometry(L):
    """Same fp16 inputs through the weights-direct and the LDS-DMA 3x3 kernels: both accumulate in fp32 over the same
    products, so they agree to fp32 summation-order noise; unsupported widths are refused loudly."""
    import proben_amd
    g = torch.Generator(device="cpu").manual_seed(29)
    x = torch.randn(2, 50, 64, 256, generator=g).cuda().half().relu()
    w = (torch.randn(256, 3, 3, 256, generator=g) / 48.0).cuda().half()
    b = torch.randn(256, generator=g).cuda()
    a = L.conv3x3_wd(x, L.conv_wd_pack(w), b, 256, relu=True).float()
    c = L.conv2d_nhwc(x, w, b, kernel=3, relu=True).float()
    torch.testing.assert_close(a, c, rtol=2e-3, atol=2e-3)
    assert not L.conv_wd_supported(3, 1, 40, 52, 256, 256)      # W % 32 != 0
    assert not L.conv_wd_supported(3, 1, 40, 64, 256, 128)      # Cout % 256 != 0
    assert not L.conv_wd_supported(1, 1, 40, 64, 256, 256)
    xx = torch.zeros(1, 8, 52, 256, device="cuda", dtype=torch.float16)
    with pytest.raises(proben_amd._lib.HipLibraryError):
        L.conv3x3_wd(xx, L.conv_wd_pack(w), b, 256)


def test_conv_transpose_detecting(L):
    """A = identity-like pixels, ASYMMETRIC weights: catches a row/col swap in the MFMA C layout."""
    Cin = Cout = 128
    x = torch.zeros(1, Cin, 4, 32).cuda().half()
    for i in range(128):
        x[0, i, i // 32, i % 32] = 1.0
    w = (torch.arange(Cout * Cin, dtype=torch.float32).reshape(Cout, Cin, 1, 1) % 97 / 97.0).cuda().half()
    out = L.conv2d_nhwc(nhwc(x), w.permute(0, 2, 3, 1).contiguous(), None, kernel=1)
    ref = torch.nn.functional.conv2d(x.float(), w.float())
    torch.testing.assert_close(out.permute(0, 3, 1, 2).float(), ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("cin", [3, 4])
def test_stem_matches_torch(L, cin):
    import proben_amd.weights as WT
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, cin, 96, 128, generator=g)
    w = torch.randn(64, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
    b = torch.randn(64, generator=g)
    x4 = torch.zeros(2, 96, 128, 4)
    x4[..., :cin] = x.permute(0, 2, 3, 1)
    out = L.conv2d_nhwc(x4.cuda().half(), WT._pack_stem(w).cuda(), b.cuda(), kernel=7, stride=2, relu=True)
    ref = torch.nn.functional.conv2d(x.cuda().half().float(), w.cuda().half().float(), b.cuda(), stride=2, padding=3).relu()
    torch.testing.assert_close(out.permute(0, 3, 1, 2).float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("cin,hw", [(3, (96, 128)), (4, (64, 96)), (3, (36, 44)), (3, (800, 1024))])
def test_fused_stem_pool_matches_torch(L, cin, hw):
    """conv7x7/2 + ReLU + max_pool2d(3,2,1) in one kernel (csrc/stem.hip) vs torch fp32 on the fp16-rounded
    operands; sizes cover ragged pooled tiles (Hp % 4 != 0, Wp % 16 != 0) and BASELINE's 800 x 1024."""
    import proben_amd.weights as WT
    g = torch.Generator().manual_seed(3)
    n = 2
    x = torch.randn(n, cin, *hw, generator=g)
    w = torch.randn(64, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
    b = torch.randn(64, generator=g)
    x4 = torch.zeros(n, *hw, 4)
    x4[..., :cin] = x.permute(0, 2, 3, 1)
    got = L.stem_conv_pool(x4.cuda().half(), WT.pack_stem_fused(w).cuda(), b.cuda())
    conv = torch.nn.functional.conv2d(x.cuda().half().float(), w.cuda().half().float(), b.cuda(), stride=2, padding=3).relu()
    ref = torch.nn.functional.max_pool2d(conv.half().float(), 3, 2, 1)
    assert got.shape == (n, hw[0] // 4, hw[1] // 4, 64)
    torch.testing.assert_close(got.permute(0, 3, 1, 2).float(), ref, rtol=4e-3, atol=4e-3)
    # and against the unfused HIP kernels (same fp16 rounding points; accumulation order differs)
    un = L.maxpool3x3s2_nhwc(L.conv2d_nhwc(x4.cuda().half(), WT._pack_stem(w).cuda(), b.cuda(), kernel=7, stride=2, relu=True))
    torch.testing.assert_close(got.float(), un.float(), rtol=2e-3, atol=2e-3)


def test_linear_matches_torch(L):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(333, 12544, generator=g).cuda().half()
    w = (torch.randn(1024, 12544, generator=g) / 112).cuda().half()
    b = torch.randn(1024, generator=g).cuda()
    got = L.linear_f16(x, w, b, relu=True).float()
    ref = torch.nn.functional.linear(x.float(), w.float(), b).relu()
    torch.testing.assert_close(got, ref, rtol=5e-3, atol=5e-3)


def test_pools_match_torch(L):
    x = torch.randn(2, 64, 37, 52).cuda().half()
    got = L.maxpool3x3s2_nhwc(nhwc(x)).permute(0, 3, 1, 2)
    assert torch.equal(got, torch.nn.functional.max_pool2d(x.float(), 3, 2, 1).half())
    got = L.subsample2_nhwc(nhwc(x)).permute(0, 3, 1, 2)
    assert torch.equal(got, torch.nn.functional.max_pool2d(x, 1, 2, 0))


def test_preprocess_pack(L):
    img = torch.rand(3, 70, 90) * 255
    dst = torch.empty(96, 96, 4, dtype=torch.float16).cuda()
    mean, std = [103.53, 116.28, 123.675], [1.0, 1.0, 1.0]
    L.preprocess_pack(img.cuda(), dst, src_kind=2, ch0=0, nch=3, flip_rgb=False, dst_hw=(70, 90), mean=mean, std=std)
    ref = torch.zeros(96, 96, 4)
    ref[:70, :90, :3] = (img - torch.tensor(mean).view(3, 1, 1)).permute(1, 2, 0)
    torch.testing.assert_close(dst.cpu().float(), ref.half().float(), rtol=0, atol=0.07)
    assert float(dst[70:].abs().max()) == 0 and float(dst[:, 90:].abs().max()) == 0
    # fused resize path: uint8 HWC -> bilinear (half-pixel) -> rounded -> normalised
    u8 = (torch.rand(48, 64, 3) * 255).to(torch.uint8)
    dst = torch.empty(96, 128, 4, dtype=torch.float16).cuda()
    L.preprocess_pack(u8.cuda(), dst, src_kind=0, ch0=0, nch=3, flip_rgb=False, dst_hw=(75, 100), mean=mean, std=std)
    up = torch.nn.functional.interpolate(u8.permute(2, 0, 1)[None].float(), size=(75, 100), mode="bilinear", align_corners=False)[0]
    ref = (up.round() - torch.tensor(mean).view(3, 1, 1)).permute(1, 2, 0)
    diff = (dst[:75, :100, :3].cpu().float() - ref).abs()
    assert float((diff > 1.01).float().mean()) < 1e-3  # rounding ties may differ by one grey level


# ------------------------------------------------------------------------------------------------ NMS
def rand_boxes(g, n, span=600.0, wh=120.0):
    xy = torch.rand(n, 2, generator=g) * span
    return torch.cat([xy, xy + torch.rand(n, 2, generator=g) * wh + 1], 1)


@pytest.mark.parametrize("name", ["rpn", "det", "uniform"])
@pytest.mark.parametrize("thr", [0.5, 0.7])
def test_nms_reproduces_the_references_own_nms(L, golden_dir, name, thr):
    """csrc/nms.hip against keep lists computed by the reference's OWN statements of horizontal greedy NMS (tests/golden/gen_nms.py:
    tests/test_nms_rotated.py:11-33 in Python and layers/csrc/nms_rotated/nms_rotated_cpu.cpp at 0 degrees, which the reference's
    tests equate with torchvision's nms / batched_nms): index-exact, through nms and through batched_nms in both dispatch modes."""
    z = np.load(os.path.join(golden_dir, "nms_reference.npz"))
    b, s = torch.from_numpy(z[name + "_boxes"]).cuda(), torch.from_numpy(z[name + "_scores"]).cuda()
    want = z[f"{name}_keep_python_{thr}"]
    np.testing.assert_array_equal(want, z[f"{name}_keep_rotated_{thr}"])
    np.testing.assert_array_equal(L.nms(b, s, thr).cpu().numpy(), want)
    # copies of the set as separate classes (up to the kernel's 16 384 boxes per image): every class must keep exactly the single-class list
    n = len(s)
    k = min(4, 16384 // n)
    bb, ss = b.repeat(k, 1), s.repeat(k)
    idx = torch.arange(k, device="cuda").repeat_interleave(n)
    keep = L.batched_nms(bb, ss, idx, thr).cpu().numpy()          # k n x 4 > 20000 elements: torchvision's per-class dispatch mode
    for c in range(k):
        mine = keep[(keep >= c * n) & (keep < (c + 1) * n)] - c * n
        np.testing.assert_array_equal(mine, want)


@pytest.mark.parametrize("case", [(48, 60, 75, 94, 4), (64, 64, 31, 47, 4), (90, 30, 135, 30, 3)])
def test_preprocess_float_resize_vs_opencv_restatement(L, case):
    """4- / 6-channel fusion inputs are floating point and go through cv2.resize in the reference; the kernel follows
    the oracle's restatement of OpenCV's INTER_LINEAR rule exactly (that restatement itself is unpinned)."""
    from oracle import resize as R
    h, w, nh, nw, c = case
    img = np.random.default_rng(h * w).integers(0, 256, size=(h, w, c)).astype(np.float32)
    ph, pw = (nh + 31) // 32 * 32, (nw + 31) // 32 * 32
    mean, std = [103.53, 116.28, 123.675, 135.438][:c], [1.0, 57.375, 2.0, 1.0][:c]
    dst = torch.empty((ph, pw, 4), dtype=torch.float16, device="cuda")
    L.preprocess_pack(torch.from_numpy(img).cuda(), dst, src_kind=1, ch0=0, nch=c, flip_rgb=False, dst_hw=(nh, nw), mean=mean, std=std)
    inv = np.float32(1.0) / np.asarray(std, dtype=np.float32)
    want = ((R.cv2_linear_resize_f64(img, nh, nw).astype(np.float32) - np.asarray(mean, dtype=np.float32)) * inv).astype(np.float16)
    assert np.array_equal(dst.cpu().numpy()[:nh, :nw, :c], want)


@pytest.mark.parametrize("case", [(2, 48, 60, 75, 94), (1, 64, 64, 31, 47), (1, 512, 640, 800, 1000), (3, 90, 30, 135, 30), (1, 100, 300, 100, 300),
                                  (1, 64, 600, 64, 100), (2, 40, 700, 100, 300)])
def test_preprocess_pil_exact_vs_oracle(L, case):
    """3-channel uint8 frames are resized exactly like Pillow (the reference's path, transform.py:92-97): every packed
    fp16 value equals (pillow_resize(img) - mean) / std computed from the oracle's restatement; padding is zero.  Upscaling and
    1:1 go through the tiled two-pass kernel (horizontal pass once per source row into LDS), vertical downscaling (more than 16
    source rows per 16 output rows) and horizontal windows above 8 taps through the per-pixel form of the same kernel."""
    from oracle import resize as R
    n, h, w, nh, nw = case
    img = np.random.default_rng(h + w).integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    img[0, : h // 3] = 255
    ph, pw = (nh + 31) // 32 * 32, (nw + 31) // 32 * 32
    mean, std = [103.53, 116.28, 123.675], [1.0, 57.375, 2.0]
    dst = torch.full((n, ph, pw, 4), 7.0, dtype=torch.float16, device="cuda")
    L.preprocess_pack_pil_u8(torch.from_numpy(img).cuda(), dst, ch0=0, nch=3, flip_rgb=False, dst_hw=(nh, nw), mean=mean, std=std)
    got = dst.cpu().numpy()
    inv = np.float32(1.0) / np.asarray(std, dtype=np.float32)
    for i in range(n):
        want = ((R.pil_bilinear_resize_u8(img[i], nh, nw).astype(np.float32) - np.asarray(mean, dtype=np.float32)) * inv).astype(np.float16)
        assert np.array_equal(got[i, :nh, :nw, :3], want), i
    assert (got[:, nh:] == 0).all() and (got[:, :, nw:] == 0).all() and (got[..., 3] == 0).all()


@pytest.mark.parametrize("n,ncls,thr", [(1, 1, 0.5), (300, 3, 0.5), (4624, 5, 0.7), (6000, 4, 0.5)])
def test_batched_nms_matches_oracle(L, n, ncls, thr):
    from oracle import nms as O
    g = torch.Generator().manual_seed(n)
    b = rand_boxes(g, n, span=200.0 if n > 1000 else 600.0)
    s = torch.rand(n, generator=g)
    if n > 14:
        s[:-1:7] = s[1::7][: len(s[:-1:7])]  # exact ties
    c = torch.randint(0, ncls, (n,), generator=g)
    got = L.batched_nms(b.cuda(), s.cuda(), c.cuda(), thr).cpu().numpy()
    want = O.batched_nms_f32(b.numpy(), s.numpy(), c.numpy(), thr, device_type="cuda")
    np.testing.assert_array_equal(got, want)


def test_nms_empty_and_float_class_ids(L):
    e = L.batched_nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), torch.zeros(0).cuda(), 0.5)
    assert e.shape == (0,) and e.dtype == torch.int64
    g = torch.Generator().manual_seed(9)
    b, s = rand_boxes(g, 50), torch.rand(50, generator=g)
    c = torch.randint(0, 3, (50,), generator=g).float()  # demo_probEn.py:57 passes torch.Tensor(classes)
    from oracle import nms as O
    got = L.batched_nms(b.cuda(), s.cuda(), c.cuda(), 0.5).cpu().numpy()
    np.testing.assert_array_equal(got, O.batched_nms_f32(b.numpy(), s.numpy(), c.numpy(), 0.5))


@pytest.mark.parametrize("case", ["rpn_like", "dead_rows", "ties", "seam_swap", "class_swap", "random", "short"])
def test_nms_presorted_input_skips_the_network_with_the_same_result(L, case):
    """csrc/nms.hip::nms_sort_kernel leaves the sorting network out when the live rows already stand in (class, score descending,
    row) order - the RPN's hand-over: every level's top-k in score order, levels in order.  Against the network (hook off): the same
    keep lists, for input that is in order (with dead rows in between, with score ties), for input that is out of order by ONE pair
    (inside a thread's stretch, across the seam of two threads' stretches, across classes), for random input and below 1024 rows."""
    from oracle import nms as O
    from proben_amd import _lib
    H = _lib.test_hooks()
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    B, per, levels = 3, 900, 5
    n = 300 if case == "short" else per * levels + 168                       # 4668 -> n_pad 8192: 8 keys per thread
    b = torch.stack([rand_boxes(g, n, span=700.0, wh=160.0) for _ in range(B)])
    s = torch.rand(B, n, generator=g)
    c = torch.zeros(B, n, dtype=torch.int32)
    valid = torch.ones(B, n, dtype=torch.uint8)
    if case != "random":
        for i in range(B):
            for l in range(levels + 1):
                lo, hi = l * per, min((l + 1) * per, n)
                if lo >= n:
                    break
                c[i, lo:hi] = l
                s[i, lo:hi] = torch.sort(s[i, lo:hi], descending=True).values
        if case == "dead_rows":
            valid = (torch.rand(B, n, generator=g) > 0.1).to(torch.uint8)
            valid[:, 0] = 0
        if case == "ties":
            s = (s * 40).round() / 40                                        # long runs of equal scores: the row index decides
        if case == "seam_swap":
            s[0, 7], s[0, 8] = s[0, 8].clone(), s[0, 7].clone()             # rows 7 | 8: two threads' stretches
            s[1, 2], s[1, 3] = s[1, 3].clone(), s[1, 2].clone()             # inside one stretch
            assert float(s[0, 8]) > float(s[0, 7]) and float(s[1, 3]) > float(s[1, 2])
        if case == "class_swap":
            c[2, per - 1], c[2, per] = 1, 0
    else:
        c = torch.randint(0, 5, (B, n), generator=g).int()
    args = (b.cuda(), s.cuda(), c.cuda(), None, valid.cuda(), 0.7, 0, 1000)
    try:
        H.pe_test_set_nms_presorted(1)
        keep1, cnt1 = L.nms_batched_raw(*args)
        H.pe_test_set_nms_presorted(0)
        keep0, cnt0 = L.nms_batched_raw(*args)
    finally:
        H.pe_test_set_nms_presorted(1)
    assert torch.equal(cnt0, cnt1) and int(cnt0.min()) > 0
    for i in range(B):
        assert torch.equal(keep0[i, : int(cnt0[i])], keep1[i, : int(cnt1[i])])
    i = B - 1                                                                 # and one image against the oracle
    live = valid[i].bool()
    want = O.batched_nms_f32(b[i][live].numpy(), s[i][live].numpy(), c[i][live].numpy(), 0.7)[:1000]
    np.testing.assert_array_equal(keep1[i, : int(cnt1[i])].cpu().numpy(), torch.nonzero(live).flatten().numpy()[want])


def test_nms_batched_raw_valid_mask_and_counts(L):
    from oracle import nms as O
    g = torch.Generator().manual_seed(21)
    B, n = 5, 700
    b = torch.stack([rand_boxes(g, n, span=150.0) for _ in range(B)])
    s = torch.rand(B, n, generator=g)
    c = torch.randint(0, 4, (B, n), generator=g).int()
    valid = (torch.rand(B, n, generator=g) > 0.2).to(torch.uint8)
    keep, cnt = L.nms_batched_raw(b.cuda(), s.cuda(), c.cuda(), None, valid.cuda(), 0.6, 0, 100)
    for i in range(B):
        sel = valid[i].bool().numpy().nonzero()[0]
        want = sel[O.batched_nms_f32(b[i].numpy()[sel], s[i].numpy()[sel], c[i].numpy()[sel], 0.6, mode="trick")][:100]
        np.testing.assert_array_equal(keep[i, : int(cnt[i])].cpu().numpy(), want)


def test_nms_per_class_fast_path_keeps_the_generic_contract(L):
    """The kernels group boxes by class (ids masked to 18 bits) and skip cross-class tiles.  Contract checks: 80 classes, a
    max_out below the number of survivors (the global top max_out by score across classes), and ids that are negative or
    collide modulo 2^18 (they share a segment but must still never suppress each other) - all against the oracle."""
    from oracle import nms as O
    g = torch.Generator().manual_seed(77)
    n = 3000
    b = rand_boxes(g, n, span=250.0)
    s = torch.rand(n, generator=g)
    c80 = torch.randint(0, 80, (n,), generator=g).int()
    keep, cnt = L.nms_batched_raw(b.cuda()[None], s.cuda()[None], c80.cuda()[None], None, None, 0.5, 0, 150)
    want = O.batched_nms_f32(b.numpy(), s.numpy(), c80.numpy(), 0.5, mode="trick")[:150]
    np.testing.assert_array_equal(keep[0, : int(cnt[0])].cpu().numpy(), want)
    ids = torch.tensor([5, 5 + (1 << 18), -3, 70000, 5 + (1 << 19)], dtype=torch.int32)
    cw = ids[torch.randint(0, 5, (n,), generator=g)]
    keep, cnt = L.nms_batched_raw(b.cuda()[None], s.cuda()[None], cw.cuda()[None], None, None, 0.5, 1, n)
    want = O.batched_nms_f32(b.numpy(), s.numpy(), cw.numpy(), 0.5, mode="vanilla")
    np.testing.assert_array_equal(keep[0, : int(cnt[0])].cpu().numpy(), want)


def test_nms_coordinate_trick_with_negative_coordinates(L):
    """torchvision's trick shifts class c by c * (max + 1): a box reaching below -1 overlaps the band of class c - 1 and the two
    can suppress each other.  The per-class organisation of the kernels must not lose that: images with a negative
    coordinate take the all-pairs route; images without stay per class; both in one batched call."""
    from oracle import nms as O
    g = torch.Generator().manual_seed(5)
    n = 600
    b = rand_boxes(g, n, span=80.0)
    b[: n // 2] -= 200.0                      # half the boxes reach far below zero: into the band of the class below
    s = torch.rand(n, generator=g)
    c = torch.randint(0, 4, (n,), generator=g).int()
    want = O.batched_nms_f32(b.numpy(), s.numpy(), c.numpy(), 0.3, mode="trick")
    per_class = O.batched_nms_f32(b.numpy(), s.numpy(), c.numpy(), 0.3, mode="vanilla")
    assert len(want) != len(per_class), "the case must actually contain cross-class suppression"
    got = L.batched_nms(b.cuda(), s.cuda(), c.cuda(), 0.3).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    b2 = torch.stack([b, b.clamp(min=0)])     # image 0 generic, image 1 per class
    keep, cnt = L.nms_batched_raw(b2.cuda(), torch.stack([s, s]).cuda(), torch.stack([c, c]).cuda(), None, None, 0.3, 0, n)
    np.testing.assert_array_equal(keep[0, : int(cnt[0])].cpu().numpy(), want)
    np.testing.assert_array_equal(keep[1, : int(cnt[1])].cpu().numpy(),
                                  O.batched_nms_f32(b2[1].numpy(), s.numpy(), c.numpy(), 0.3, mode="trick"))


# ------------------------------------------------------------------------------------------------ RPN
@pytest.mark.parametrize("two_stage", [False, True])
def test_rpn_select_matches_oracle(two_stage):
    import ctypes
    import proben_amd  # noqa: F401
    from oracle import detector as D
    from proben_amd import _lib
    from proben_amd.rcnn import SCALE_CLAMP, cell_anchor_table
    spec = D.DetectorSpec()
    g = torch.Generator().manual_seed(77)
    N, shapes, strides = 2, ([(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)] if two_stage else [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)]), [4, 8, 16, 32, 64]
    heads, lg_l, dl_l = [], [], []
    for (h, w) in shapes:
        hd = torch.randn(N, h, w, 16, generator=g)
        hd[..., 3:15] *= 0.5
        hd[0, 0, 0, 0] = float("nan")
        hd[1, 1, 1, 5] = float("inf")
        hd[:, 2, :, 1] = 0.25  # exact ties
        heads.append(hd)
        lg_l.append(hd[..., :3].permute(0, 3, 1, 2).contiguous())                       # [N, A, H, W]
        dl_l.append(hd[..., 3:15].permute(0, 3, 1, 2).contiguous())                     # [N, 4A, H, W]
    sizes = [(390, 500), (400, 512)] if two_stage else [(150, 200), (160, 208)]
    want = D.select_proposals(lg_l, dl_l, strides, sizes, spec)
    # --- HIP: selection kernel + NMS + gather
    from proben_amd import layers as L
    hd_dev = [h.cuda().contiguous() for h in heads]
    topk = [min(1000, h * w * 3) for h, w in shapes]
    ncand = sum(topk)
    cb = torch.empty(N, ncand, 4).cuda(); cs = torch.empty(N, ncand).cuda()
    cl = torch.empty(N, ncand, dtype=torch.int32).cuda(); cv = torch.empty(N, ncand, dtype=torch.uint8).cuda()
    ptrs = (ctypes.c_void_p * 5)(*[h.data_ptr() for h in hd_dev])
    hw = (ctypes.c_int32 * 10)(*sum([list(s) for s in shapes], []))
    cells = (ctypes.c_float * 60)(*cell_anchor_table(spec.anchor_sizes, spec.aspect_ratios))
    sz = torch.tensor(sizes, dtype=torch.int32).cuda()
    sbytes = _lib.lib().pe_rpn_scratch_bytes(hw, 5, N) if two_stage else 0
    assert (sbytes > 0) == two_stage
    scratch = torch.empty(max(sbytes, 8), dtype=torch.uint8).cuda()
    st = _lib.lib().pe_rpn_select_topk(ptrs, hw, (ctypes.c_int32 * 5)(*strides), cells, 5, N, 16, 1000, _lib.ptr(sz),
                                      SCALE_CLAMP, _lib.ptr(cb), _lib.ptr(cs), _lib.ptr(cl), _lib.ptr(cv), ncand,
                                      _lib.ptr(scratch) if two_stage else None, sbytes, _lib.stream())
    _lib.check(st, "rpn")
    keep, cnt = L.nms_batched_raw(cb, cs, cl, None, cv, 0.7, 0, 1000)
    for n in range(N):
        k = keep[n, : int(cnt[n])].long()
        gb, gs = cb[n][k].cpu(), cs[n][k].cpu()
        wb, ws = want[n]
        assert gb.shape == wb.shape
        np.testing.assert_array_equal(gs.numpy(), ws.numpy())
        np.testing.assert_allclose(gb.numpy(), wb.numpy(), rtol=1e-5, atol=1e-4)  # expf: device vs libm


# ------------------------------------------------------------------------------------------------ ROIAlign
def test_roi_align_fp32_bit_exact_vs_oracle(L, golden_dir):
    """The reference's own pooler fixture (4 FPN levels, level assignment, empty + full-frame boxes)."""
    z = np.load(os.path.join(golden_dir, "detector_ops.npz"))
    feats = [torch.from_numpy(z[f"pool_feat_l{i}"]) for i in range(4)]
    boxes = torch.stack([torch.from_numpy(z["pool_boxes_0"]), torch.from_numpy(z["pool_boxes_1"])])  # [2,40,4]
    out, lv = L.roi_align_nhwc([nhwc(f).cuda() for f in feats], boxes.cuda(), scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32],
                               pooled=(7, 7), counts=None, per_image=40, want_levels=True)
    got = out.permute(0, 3, 1, 2).cpu().numpy()
    np.testing.assert_array_equal(got, z["pool_out"])


def test_roi_align_module_matches_reference_tables(L):
    inp = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5).cuda()
    rois = torch.tensor([[0, 1, 1, 3, 3.0]]).cuda()
    old = [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]]
    new = [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]]
    np.testing.assert_allclose(L.ROIAlign((4, 4), 1.0, 0, aligned=False)(inp, rois)[0, 0].cpu(), old)
    np.testing.assert_allclose(L.ROIAlign((4, 4), 1.0, 0, aligned=True)(inp, rois)[0, 0].cpu(), new)
    # empty box -> zeros, empty batch -> (0, C, 7, 7)   (tests/test_roi_align.py:94-111)
    x = torch.rand(1, 3, 9, 9).cuda()
    assert float(L.ROIAlign(7, 1.0, 0)(x, torch.tensor([[0, 3, 3, 3, 3.0]]).cuda()).abs().max()) == 0.0
    assert L.ROIAlign(7, 1.0, 0)(x, torch.zeros(0, 5).cuda()).shape == (0, 3, 7, 7)


def test_roi_align_fp16_and_dead_rows(L):
    from oracle import roi_align as RA
    g = torch.Generator().manual_seed(4)
    f = torch.randn(2, 256, 50, 64, generator=g).half()
    boxes = torch.stack([rand_boxes(g, 30, span=600, wh=300) for _ in range(2)])
    counts = torch.tensor([30, 11], dtype=torch.int32)
    out = L.roi_align_nhwc([nhwc(f).cuda()], boxes.cuda(), scales=[1 / 16], pooled=(7, 7), counts=counts.cuda(), per_image=30)
    rois = torch.cat([torch.cat([torch.full((30, 1), float(i)), boxes[i]], 1) for i in range(2)])
    ref = RA.roi_align_forward(f.float(), rois, 1 / 16, 7, 7, 0, True)
    got = out.permute(0, 3, 1, 2).float().cpu()
    torch.testing.assert_close(got[:41], ref[:41], rtol=2e-3, atol=2e-3)
    assert float(got[41:].abs().max()) == 0.0


@pytest.mark.parametrize("N,P", [(3, 1000), (2, 37), (1, 2048), (2, 2100)])
def test_roi_align_sorted_order_moves_no_result(L, N, P):
    """pe_roi_align_nhwc_sorted (level / Morton processing order, XCD-contiguous) against pe_roi_align_nhwc: the same bits at
    the same places, the same levels, dead rows zero; per_image > 2048 falls back to the given order."""
    g = torch.Generator().manual_seed(N * 7919 + P)
    feats = [torch.randn(N, 200 >> l, 256 >> l, 64, generator=g).half().cuda() for l in range(4)]
    ctr = torch.rand(N, P, 2, generator=g) * torch.tensor([1000.0, 780.0])
    wh = torch.exp(torch.rand(N, P, 2, generator=g) * 5.5 + 1.5)            # 4 .. 1100 px: every level
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=2).cuda()
    counts = torch.tensor([P, P // 3, 0][:N] if N > 1 else [P - 5], dtype=torch.int32).cuda()
    kw = dict(scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], pooled=(7, 7), counts=counts, per_image=P, want_levels=True)
    a, la = L.roi_align_nhwc(feats, boxes, sort=False, **kw)
    b, lb = L.roi_align_nhwc(feats, boxes, sort=True, **kw)
    assert torch.equal(la, lb) and set(la.unique().tolist()) >= {0, 1, 2, 3}
    assert torch.equal(a, b)
    assert float(a.view(N, P, -1)[0, :counts[0]].abs().sum()) > 0
    if N > 1:
        assert float(a.view(N, P, -1)[1, counts[1]:].abs().sum()) == 0


@pytest.mark.parametrize("pooled,ratio,aligned,tiny", [((7, 7), 0, True, False), ((7, 7), 0, True, True), ((8, 8), 0, False, False), ((3, 5), 2, True, False), ((7, 7), 3, False, False),
                                                       ((7, 7), 0, True, "nonfinite"), ((8, 8), 0, False, "nonfinite")])
def test_roi_align_fast_form_is_bit_identical(L, pooled, ratio, aligned, tiny):
    """The wave-uniform form (fp16, C = 256: two bins per wave walk the wider of their two windows, scalar pixel walk, v_fma_mix,
    three-instruction exact division) against the per-lane form it replaces in the detector: the same bits.  Boxes cover every
    level, sub-pixel and whole-image sizes, boxes partly and wholly outside the image (empty windows next to live ones), the last
    image's bottom-right corner (padded columns run past the level's end: bounded buffer loads) and - single level, stride 4 - bins
    above the table size (tap-form fallback inside the fast kernel); `tiny` scales the features into fp16 subnormals; "nonfinite"
    (round 6) sprinkles +-Inf and NaN pixels over every level, the last pixel of the last image included: a bin next to such
    a pixel stays finite in the per-lane form and in the reference (it never samples it), and so it must in the wave-uniform form - the
    padded columns of the narrower bin are not loaded - while a bin that does sample it carries the same non-finite bits.  (Adaptive
    sampling ratio only - the detector's: with a FIXED ratio whose samples lie more than a pixel apart the table's window holds
    unsampled pixels at weight zero, and a non-finite one among them poisons the bin in both table forms; csrc/roi_align.hip says so.)"""
    from proben_amd import _lib
    H = _lib.test_hooks()
    N, P = 3, 700
    nonfinite = tiny == "nonfinite"
    tiny = tiny is True
    g = torch.Generator().manual_seed(pooled[0] * 131 + ratio * 17 + int(aligned) + 2 * int(tiny) + 5 * int(nonfinite))
    feats = [(torch.randn(N, 200 >> l, 256 >> l, 256, generator=g) * (2e-6 if tiny else 1.0)).half().cuda() for l in range(4)]
    feats[2][:, 5:9, 3:11] = 0
    if nonfinite:
        for l, f in enumerate(feats):
            n_bad = 400 >> l
            iy = torch.randint(0, f.shape[1], (n_bad,), generator=g)
            ix = torch.randint(0, f.shape[2], (n_bad,), generator=g)
            ic = torch.randint(0, 256, (n_bad,), generator=g)
            im = torch.randint(0, N, (n_bad,), generator=g)
            vals = torch.tensor([float("inf"), float("-inf"), float("nan")])[torch.randint(0, 3, (n_bad,), generator=g)].half().cuda()
            f[im.cuda(), iy.cuda(), ix.cuda(), ic.cuda()] = vals
            f[N - 1, -1, -1, :] = float("inf")       # the level's last pixel
            f[:, :, -1, 7] = float("nan")           # a whole last column of one channel: every right-edge bin's neighbour
            f[:, 0, 0, :] = 1.0                     # pixel (0, 0) stays finite: the reference's CPU kernel gives an out-of-image SAMPLE weight 0 at
                                                    # position 0 (ROIAlign_cpu.cpp:60-70) and multiplies - its CUDA twin skips the sample, as this kernel does
    ctr = torch.rand(N, P, 2, generator=g) * torch.tensor([1100.0, 860.0]) - torch.tensor([40.0, 30.0])
    wh = torch.exp(torch.rand(N, P, 2, generator=g) * 7.0 - 0.5)              # 0.6 .. 660 px, independent sides: elongated boxes too
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=2)
    boxes[:, 0] = torch.tensor([-50.0, -40.0, 1100.0, 900.0])                # beyond the image on every side
    boxes[:, 1] = torch.tensor([1500.0, 900.0, 1600.0, 1000.0])              # wholly outside: empty windows
    boxes[:, 2] = torch.tensor([1000.0, 780.0, 1024.0, 800.0])               # bottom-right corner
    boxes[:, 3] = torch.tensor([1010.0, 100.0, 1300.0, 300.0])               # left columns live, right columns outside
    boxes[:, 4] = torch.tensor([100.0, 100.0, 100.0, 100.0])                 # empty box
    boxes = boxes.cuda()
    counts = torch.tensor([P, P - 9, 5], dtype=torch.int32).cuda()
    cases = [dict(feats=feats, scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32]), dict(feats=feats[:1], scales=[1 / 4]), dict(feats=feats[3:], scales=[1 / 32])]
    try:
        for case in cases:
            kw = dict(scales=case["scales"], pooled=pooled, sampling_ratio=ratio, aligned=aligned, counts=counts, per_image=P)
            outs = []
            for fast in (1, 0):
                H.pe_test_set_roi_fast(fast)
                outs.append([L.roi_align_nhwc(case["feats"], boxes, sort=s, **kw) for s in (False, True)])
            def why(x, y):      # where two results differ: how many elements, and of which kind
                d = x.view(torch.int16) != y.view(torch.int16)
                xf, yf = x.float()[d], y.float()[d]
                return {"differing": int(d.sum()), "both_nan": int((torch.isnan(xf) & torch.isnan(yf)).sum()), "fast_finite_other_not": int((torch.isfinite(xf) & ~torch.isfinite(yf)).sum()),
                        "other_finite_fast_not": int((~torch.isfinite(xf) & torch.isfinite(yf)).sum()), "inf_vs_nan": int((torch.isinf(xf) != torch.isinf(yf)).sum()),
                        "first": (xf[:4].tolist(), yf[:4].tolist(), x.view(torch.int16)[d][:4].tolist(), y.view(torch.int16)[d][:4].tolist())}
            assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)), why(outs[0][0], outs[1][0])
            assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16)), why(outs[0][1], outs[1][1])
            if nonfinite:
                bad = ~torch.isfinite(outs[0][0].float())
                assert 0 < int(bad.sum()) < bad.numel() // 4      # some bins sample a poisoned pixel, most do not
                if len(case["feats"]) == 1 and len(case["scales"]) == 1 and case["scales"][0] == 1 / 4:
                    # which bins those are is the reference's business: the oracle (restatement of ROIAlign_cpu.cpp) on the same fp16
                    # features in fp32 must be non-finite in exactly the same (roi, bin, channel) places - a window's padding, a
                    # neighbour bin's column or what lies behind the level never shows
                    from oracle import roi_align as RA
                    f = case["feats"][0].float().cpu().permute(0, 3, 1, 2).contiguous()
                    cnt = counts.cpu().tolist()
                    rois = torch.cat([torch.cat([torch.full((cnt[i], 1), float(i)), boxes[i, :cnt[i]].cpu()], 1) for i in range(N)])
                    ref = RA.roi_align_forward(f, rois, 1 / 4, pooled[0], pooled[1], ratio, aligned).permute(0, 2, 3, 1)
                    got = torch.cat([outs[0][0].view(N, P, pooled[0], pooled[1], 256)[i, :cnt[i]] for i in range(N)]).float().cpu()
                    fg, fr = torch.isfinite(got), torch.isfinite(ref)
                    assert torch.equal(fg, fr), {"hip_only_nonfinite": int((~fg & fr).sum()), "oracle_only_nonfinite": int((fg & ~fr).sum()), "where": (~fg & fr).nonzero()[:5].tolist() + (fg & ~fr).nonzero()[:5].tolist()}
                    # (WHICH non-finite value a poisoned bin holds may differ: the reference multiplies every sample's four taps one by
                    # one - a zero-weight tap on an Inf pixel makes NaN -, the table form multiplies a pixel's SUMMED weight once - Inf)
            else:
                assert float(outs[0][0].float().abs().sum()) > 0
    finally:
        H.pe_test_set_roi_fast(1)


def test_roi_align_backward_matches_oracle_and_autograd(L):
    """Training half: the backward kernel == the oracle's restatement of ROIAlign_cpu.cpp:221-394 (atomics:
    tolerance, not bits), rows beyond counts contribute nothing, and `layers.ROIAlign` is differentiable end to end."""
    from oracle import roi_align as RA
    g = torch.Generator().manual_seed(8)
    N, C, H, W, P = 2, 16, 25, 32, 20
    boxes = torch.stack([rand_boxes(g, P, span=420, wh=200) for _ in range(N)])
    counts = torch.tensor([P, 7], dtype=torch.int32)
    go = torch.randn(N * P, 7, 7, C, generator=g)
    for dt, tol in ((torch.float32, 5e-5), (torch.float16, 2e-3)):   # fp32: summation order differs (atomics vs the oracle's sequential adds)
        (gin,) = L.roi_align_backward_nhwc(go.to(dt).cuda(), boxes.cuda(), [(N, H, W, C)], scales=[1 / 16], counts=counts.cuda(), per_image=P)
        rois = torch.cat([torch.cat([torch.full((int(counts[i]), 1), float(i)), boxes[i, : int(counts[i])]], 1) for i in range(N)])
        live = torch.cat([go[i * P: i * P + int(counts[i])] for i in range(N)]).to(dt).float()
        ref = RA.roi_align_backward(live.permute(0, 3, 1, 2), rois, 1 / 16, 7, 7, N, C, H, W, 0, True)
        torch.testing.assert_close(gin.permute(0, 3, 1, 2).cpu(), ref, rtol=tol, atol=tol)
    # four FPN levels in one launch: the same rule assigns the level in forward and backward
    shapes = [(N, 200 // s, 256 // s, 8) for s in (1, 2, 4, 8)]
    big = torch.stack([rand_boxes(g, P, span=700, wh=500) for _ in range(N)])
    go4 = torch.randn(N * P, 7, 7, 8, generator=g)
    grads = L.roi_align_backward_nhwc(go4.cuda(), big.cuda(), shapes, scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], per_image=P)
    feats = [torch.zeros(sh, device="cuda") for sh in shapes]
    _, lv = L.roi_align_nhwc(feats, big.cuda(), scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], pooled=(7, 7), per_image=P, want_levels=True)
    rois = torch.cat([torch.cat([torch.full((P, 1), float(i)), big[i]], 1) for i in range(N)])
    for l, sc in enumerate((1 / 4, 1 / 8, 1 / 16, 1 / 32)):
        sel = (lv.cpu() == l).nonzero().squeeze(1)
        ref = RA.roi_align_backward(go4[sel].permute(0, 3, 1, 2), rois[sel], sc, 7, 7, N, 8, shapes[l][1], shapes[l][2], 0, True)
        torch.testing.assert_close(grads[l].permute(0, 3, 1, 2).cpu(), ref, rtol=5e-5, atol=5e-5)
    # autograd through the module (layers/roi_align.py:10-49)
    x = torch.randn(1, 3, 12, 14, generator=g).cuda().requires_grad_()
    r = torch.tensor([[0, 1.0, 2.0, 11.0, 9.0], [0, 0.0, 0.0, 5.5, 5.5]]).cuda()
    y = L.ROIAlign((3, 3), 0.5, 2, aligned=True)(x, r)
    w = torch.randn(y.shape, generator=g).cuda()
    (y * w).sum().backward()
    ref = RA.roi_align_backward(w.cpu(), r.cpu(), 0.5, 3, 3, 1, 3, 12, 14, 2, True)
    torch.testing.assert_close(x.grad.cpu(), ref, rtol=5e-5, atol=5e-5)


# ------------------------------------------------------------------------------------------------ box head
def test_boxhead_matches_oracle_quirks():
    import ctypes
    import proben_amd  # noqa: F401
    from oracle import detector as D
    from proben_amd import _lib
    from proben_amd import layers as L
    from proben_amd.rcnn import SCALE_CLAMP
    g = torch.Generator().manual_seed(31)
    N, P, K = 3, 1000, 3
    stride = 24
    head = torch.zeros(N, P, stride)
    head[..., : K + 1] = torch.randn(N, P, K + 1, generator=g) * 2.5
    head[..., K + 1: 5 * K + 1] = torch.randn(N, P, 4 * K, generator=g)
    head[..., 5 * K + 1] = torch.randn(N, P, generator=g) * 0.5
    head[1, 17, K + 2] = float("nan")   # Q4: non-finite row dropped from boxes/scores only
    head[2, 5, 0] = float("inf")
    props = torch.stack([rand_boxes(g, P, span=800, wh=250) for _ in range(N)])
    pcnt = torch.tensor([1000, 640, 1000], dtype=torch.int32)
    sizes = [(800, 1000), (768, 960), (800, 1000)]
    outs = [(512, 640), (492, 614), (512, 640)]
    spec = D.DetectorSpec()
    dev = "cuda"
    cmax = P * K
    hd, pr = head.view(N * P, stride).cuda(), props.cuda()
    cb = torch.empty(N, cmax, 4, device=dev); cs = torch.empty(N, cmax, device=dev)
    cc = torch.empty(N, cmax, dtype=torch.int32, device=dev); cr = torch.empty(N, cmax, 2, dtype=torch.int32, device=dev)
    ccnt = torch.empty(N, dtype=torch.int32, device=dev); ctot = torch.empty(N, dtype=torch.int32, device=dev); probs = torch.empty(N, P, K + 1, device=dev)
    sz = torch.tensor(sizes, dtype=torch.int32, device=dev); osz = torch.tensor(outs, dtype=torch.int32, device=dev)
    lib = _lib.lib()
    _lib.check(lib.pe_boxhead_candidates(_lib.ptr(hd), stride, N, P, K, _lib.ptr(pcnt.cuda()), _lib.ptr(pr), _lib.ptr(sz),
                                         (ctypes.c_float * 4)(10, 10, 5, 5), SCALE_CLAMP, 0.5, cmax, _lib.ptr(cb), _lib.ptr(cs),
                                         _lib.ptr(cc), _lib.ptr(cr), _lib.ptr(ccnt), _lib.ptr(ctot), _lib.ptr(probs), _lib.stream()), "cand")
    assert torch.equal(ctot, ccnt)                       # no overflow here: the uncapped total equals the kept count
    # a cap smaller than the candidates is REPORTED: kept count = cap, total = the real number
    small = 3
    sb_, ss_, sc_, sr_ = cb[:, :small].contiguous(), cs[:, :small].contiguous(), cc[:, :small].contiguous(), cr[:, :small].contiguous()
    c2, t2 = torch.empty_like(ccnt), torch.empty_like(ctot)
    _lib.check(lib.pe_boxhead_candidates(_lib.ptr(hd), stride, N, P, K, _lib.ptr(pcnt.cuda()), _lib.ptr(pr), _lib.ptr(sz),
                                         (ctypes.c_float * 4)(10, 10, 5, 5), SCALE_CLAMP, 0.5, small, _lib.ptr(sb_), _lib.ptr(ss_),
                                         _lib.ptr(sc_), _lib.ptr(sr_), _lib.ptr(c2), _lib.ptr(t2), _lib.ptr(torch.empty_like(probs)),
                                         _lib.stream()), "cand-small")
    assert torch.equal(t2, ctot) and torch.equal(c2, ctot.clamp(max=small))
    keep, kcnt = L.nms_batched_raw(cb, cs, cc, ccnt, None, 0.5, 0, 100)
    D_ = 100
    o = {k: torch.empty(s, dtype=t, device=dev) for k, s, t in [
        ("boxes", (N, D_, 4), torch.float32), ("scores", (N, D_), torch.float32), ("classes", (N, D_), torch.int32),
        ("logits", (N, D_, K + 1), torch.float32), ("probs", (N, D_, K), torch.float32), ("vars", (N, D_), torch.float32),
        ("rows", (N, D_), torch.int32), ("counts", (N,), torch.int32)]}
    _lib.check(lib.pe_boxhead_finalize(_lib.ptr(hd), stride, N, P, K, cmax, D_, 0, _lib.ptr(probs), _lib.ptr(cb), _lib.ptr(cs),
                                       _lib.ptr(cc), _lib.ptr(cr), _lib.ptr(keep), _lib.ptr(kcnt), _lib.ptr(sz), _lib.ptr(osz),
                                       _lib.ptr(o["boxes"]), _lib.ptr(o["scores"]), _lib.ptr(o["classes"]), _lib.ptr(o["logits"]),
                                       _lib.ptr(o["probs"]), _lib.ptr(o["vars"]), _lib.ptr(o["rows"]), _lib.ptr(o["counts"]),
                                       _lib.stream()), "final")
    for n in range(N):
        r = int(pcnt[n])
        h = head[n, :r]
        det = D.select_detections(h[:, : K + 1], h[:, K + 1: 5 * K + 1], torch.exp(h[:, 5 * K + 1: 5 * K + 2]),
                                  props[n, :r], sizes[n], spec)
        want = D.postprocess(det, sizes[n], outs[n])
        c = int(o["counts"][n])
        assert c == len(want["boxes"]), (n, c, len(want["boxes"]))
        np.testing.assert_array_equal(o["classes"][n, :c].cpu().numpy(), want["classes"].numpy())
        if n == 0:  # with dropped (non-finite) rows the oracle reports post-filter row ids, the kernel original ones
            np.testing.assert_array_equal(o["rows"][n, :c].cpu().numpy(), want["roi_index"].numpy())
        np.testing.assert_allclose(o["scores"][n, :c].cpu().numpy(), want["scores"].numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(o["boxes"][n, :c].cpu().numpy(), want["boxes"].numpy(), rtol=1e-5, atol=2e-4)
        np.testing.assert_array_equal(o["logits"][n, :c].cpu().numpy(), want["class_logits"].numpy())
        np.testing.assert_allclose(o["probs"][n, :c].cpu().numpy(), want["prob_score"].numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(o["vars"][n, :c].cpu().numpy(), want["vars"].numpy().reshape(-1), rtol=2e-6)
