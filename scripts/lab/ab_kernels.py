"""Lab only: the same handful of backbone launches (batch 32) timed on several builds of the library in ONE call on one box.
    python scripts/lab/ab_kernels.py [--only tail,ring,b64] product exp_<tag> exp_<tag2> ...
Each library gets its own process (the library is loaded once per process); `product` = libproben_hip.so, `exp_x` = libproben_hip_exp_x.so
(scripts/lab/build_variant.py).  Rounds alternate between the libraries so that clock / temperature drift shows as spread, not as a winner.
Every variant's outputs are compared with the product's (max abs difference; 0 = same bits)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(lib_name, only, dump):
    import torch
    import proben_amd  # noqa: F401
    from proben_amd import _lib, layers as L
    if lib_name != "product":
        _lib.LIB_PATH = _lib.LIB_PATH.replace(".so", "_%s.so" % lib_name)
        assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH

    def timed(fn, reps=40):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    torch.manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda")
    cases = {}
    if "tail" in only:
        N, H, W, C, CT = 32, 50, 64, 256, 1024
        x = rnd(N, H, W, C).half().relu()
        w2 = (rnd(C, 3, 3, C) / (C * 9) ** 0.5).half(); b2 = rnd(C) * 0.1
        w3 = (rnd(CT, C) / C ** 0.5).half(); b3 = rnd(CT) * 0.1
        res = rnd(N, H, W, CT).half().relu()
        out = torch.empty(N, H, W, CT, device="cuda", dtype=torch.float16)
        pk2, pk3 = L.conv_wd_pack(w2), L.conv_wd_pack_tail(w3)
        cases["tail res4"] = ((lambda x=x, pk2=pk2, b2=b2, pk3=pk3, b3=b3, res=res, CT=CT, out=out: L.bottleneck_tail_wd(x, pk2, b2, pk3, b3, res, CT, out=out)), out)
    if "ring" in only:
        for name, (N, H, W, K, Co, rm) in {"ring res4 conv1": (32, 50, 64, 1024, 256, 0), "ring res3 conv3": (32, 100, 128, 128, 512, 1),
                                           "ring res5 conv3": (32, 25, 32, 512, 2048, 1), "ring fc1": (1, 1, 32000, 12544, 1024, 0)}.items():
            x = rnd(N, H, W, K).half().relu()
            w = (rnd(Co, 1, 1, K) / K ** 0.5).half(); b = rnd(Co) * 0.1
            r = rnd(N, H, W, Co).half().relu() if rm else None
            o = torch.empty(N, H, W, Co, device="cuda", dtype=torch.float16)
            cases[name] = ((lambda x=x, w=w, b=b, r=r, o=o, rm=rm: L.conv2d_nhwc(x, w, b, kernel=1, relu=True, residual=r, residual_mode=rm, out=o)), o)
    if "b64" in only:
        N, H, W = 32, 200, 256
        t1 = rnd(N, H, W, 64).half().relu(); x = rnd(N, H, W, 256).half().relu(); s = rnd(N, H, W, 64).half().relu()
        w2 = (rnd(64, 3, 3, 64) / 24).half(); b2 = rnd(64) * 0.1
        w3 = (rnd(256, 64) / 8).half(); b3 = rnd(256) * 0.1
        wsc = (rnd(256, 64) / 8).half(); bsc = rnd(256) * 0.1
        w1n = (rnd(64, 256) / 16).half(); b1n = rnd(64) * 0.1
        out = torch.empty(N, H, W, 256, device="cuda", dtype=torch.float16)
        t1n = torch.empty(N, H, W, 64, device="cuda", dtype=torch.float16)
        for sc, nxt in ((True, True), (False, True), (False, False)):
            pk = L.bneck64_pack(w2, w3, wsc if sc else None, w1n if nxt else None)
            cases["b64 sc%d next%d" % (sc, nxt)] = ((lambda sc=sc, nxt=nxt, pk=pk, t1=t1, s=s, x=x, b2=b2, b3=b3, out=out, t1n=t1n: L.bneck64(t1, s if sc else x, pk, b2, b3, bsc if sc else None, b1n if nxt else None,
                                                                                            out=out, t1_next=t1n if nxt else None)), out)
    if "stem" in only:
        pass
    import hashlib
    for name, (fn, o) in cases.items():
        t = [timed(fn) for _ in range(3)]
        fn(); torch.cuda.synchronize()
        h = hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]
        print("RESULT\t%s\t%s\t%s\t%s" % (lib_name, name, " ".join("%.1f" % v for v in t), h), flush=True)


def main():
    args = sys.argv[1:]
    only = "tail,ring,b64"
    if args and args[0] == "--only":
        only, args = args[1], args[2:]
    if args and args[0] == "--worker":
        return worker(args[1], only.split(","), None)
    libs = args or ["product"]
    rows = {}
    for rnd_i in range(2):
        for lib in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--only", only, "--worker", lib], capture_output=True, text=True)
            if out.returncode:
                print(lib, "FAILED", out.stderr[-2000:])
                continue
            for line in out.stdout.splitlines():
                if line.startswith("RESULT"):
                    _, l, name, t, h = line.split("\t")
                    rows.setdefault(name, {}).setdefault(l, []).append((t, h))
    for name, per in rows.items():
        print(name)
        ref = per.get("product", [(None, None)])[0][1]
        for l, v in per.items():
            print("   %-22s us: %s   %s" % (l, " | ".join(t for t, _ in v), "same bits" if v[0][1] == ref else "BITS DIFFER" if ref else v[0][1]))


if __name__ == "__main__":
    main()
