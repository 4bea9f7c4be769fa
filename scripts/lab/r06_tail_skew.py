"""Round 6: does de-phasing the fused res4 tail's workgroups pay?  All 512 first-round workgroups start together and change from the
MFMA-bound 3x3 phase to the HBM-bound conv3 phase together; the hook delays one half of the chip (by XCD / CU / shader engine).
    python scripts/lab/r06_tail_skew.py          (GPU box)
Result (profiles/r06_tail_skew.txt): no gain (best -1 %, odd CUs 20-35 us late; XCD- and SE-level skews are slower), so the hook is NOT in
the tree.  To repeat: in conv_wd.h add `int skew_cycles, skew_mode, skew_first;` to ConvWdArgs and, at the top of conv3x3_wd_kernel,
    if constexpr (HEAD == 2) if (a.skew_cycles > 0 && (int)blockIdx.x < a.skew_first) {
        bool late = a.skew_mode == 1 ? (blockIdx.x & 4) != 0
                  : a.skew_mode == 2 ? (__builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4) & 1) != 0      // HW_ID.CU_ID bit 0
                  :                    (__builtin_amdgcn_s_getreg((2 << 11) | (13 << 6) | 4) & 1) != 0;    // HW_ID.SE_ID bit 0
        if (late) { const long long until = (long long)__builtin_readcyclecounter() + a.skew_cycles;
                    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(32); } }
and in conv_wd.hip a hook `pe_test_set_tail_skew(cycles, mode)` that fills the three fields (skew_first = 512) in pe_bottleneck_tail_wd_f16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proben_amd  # noqa: E402,F401
from proben_amd import _lib, layers as L  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    hooks = _lib.test_hooks()
    N, H, W, C, CT = 32, 50, 64, 256, 1024
    torch.manual_seed(0)
    x = torch.randn(N, H, W, C, device="cuda").half().relu()
    w2 = (torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5).half()
    b2 = torch.randn(C, device="cuda") * 0.1
    w3 = (torch.randn(CT, 1, 1, C, device="cuda") / C ** 0.5).half()
    b3 = torch.randn(CT, device="cuda") * 0.1
    res = torch.randn(N, H, W, CT, device="cuda").half().relu()
    out = torch.empty(N, H, W, CT, device="cuda", dtype=torch.float16)
    pk2 = L.conv_wd_pack(w2)
    pk3 = L.conv_wd_pack_tail(w3.reshape(CT, C))
    fn = lambda: L.bottleneck_tail_wd(x, pk2, b2, pk3, b3, res, CT, out=out)
    fn()
    ref = out.clone()
    for rep in range(2):
        for mode in (0, 1, 2, 3):
            for cyc in ((0,) if mode == 0 else (20000, 40000, 70000, 100000, 140000, 200000)):
                hooks.pe_test_set_tail_skew(cyc, mode)
                ms = timed(fn)
                same = bool(torch.equal(out, ref))
                print(f"mode {mode} skew {cyc:7d} cycles: {ms * 1e3:7.1f} us  bits {'same' if same else 'DIFFER'}", flush=True)
    hooks.pe_test_set_tail_skew(0, 0)


if __name__ == "__main__":
    main()
