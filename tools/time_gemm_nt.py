"""The three trunk products of the large-minibatch PPO step (csrc/sdx_gemm_nt.h) timed alone, per layer, as the step launches them (the
same layer of the three networks per launch), against torch matmul for the result and against the MFMA peak for the rate.

    python tools/time_gemm_nt.py [--mb 32768] [--bf16] [--out profiles/r4_bigmb_products_<dtype>_mb<MB>.txt]
"""
import argparse
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from seqdex_amd import _abi  # noqa: E402


class NtArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("kchunk", C.c_int), ("Cf", C.c_void_p), ("ldc", C.c_int), ("cz", C.c_size_t), ("Cn", C.c_void_p), ("ldn", C.c_int),
                ("Ct", C.c_void_p), ("ldt", C.c_int), ("bias", C.c_void_p), ("H", C.c_void_p), ("ldh", C.c_int), ("Ht", C.c_void_p), ("ldht", C.c_int), ("rowsum", C.c_void_p)]


EPI_FWD, EPI_NN, EPI_TN = 1, 3, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=32768)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--ablate", action="store_true", help="also time the L1 forward / data-gradient products with some or all of their output copies switched off (what the epilogue costs)")
    a = ap.parse_args()
    lib = _abi.load_library()
    lib.sdxpk_gemm_nt_launch.restype = C.c_int
    lib.sdxpk_gemm_nt_launch.argtypes = [C.c_int, C.c_int, C.POINTER(NtArgs), C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.bf16 else torch.float32
    KC = 64 if a.bf16 else 32
    MB = a.mb
    peak = 2500.0 if a.bf16 else 157.3
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
    pad = lambda k: (k + KC - 1) // KC * KC
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lines = []

    def run(name, epi, args3, flops, splits=1, check=None):
        arr = (NtArgs * 3)(*args3)
        for _ in range(3):
            assert lib.sdxpk_gemm_nt_launch(int(a.bf16), epi, arr, 3, splits, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            lib.sdxpk_gemm_nt_launch(int(a.bf16), epi, arr, 3, splits, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        tf = flops / us / 1e6
        err = check() if check else float("nan")
        ln = "%-46s %9.1f us %8.1f TFLOP/s %5.1f %% of %s peak   max rel err vs torch %.2e" % (name, us, tf, 100 * tf / peak, "bf16" if a.bf16 else "fp32", err)
        lines.append(ln)
        print(ln, flush=True)
        return us

    units = [1024, 512, 256]
    ins = [(396, 396, 564), (1024,) * 3, (512,) * 3]
    total_us, total_fl = 0.0, 0.0
    for l in range(3):
        N = units[l]
        # ---- forward: H = ELU(X W^T + b)
        Xs, Ws, Hs, Hn, Ht, bs = [], [], [], [], [], []
        args = []
        for net in range(3):
            K = ins[l][net]; Kp = pad(K)
            X = torch.zeros(MB, Kp, device=dev); X[:, :K] = rnd(MB, K)
            W = torch.zeros(N, Kp, device=dev); W[:, :K] = rnd(N, K) / K ** 0.5
            Xs.append(X.to(dt).contiguous()); Ws.append(W.to(dt).contiguous())
            Hs.append(torch.empty(MB, N, device=dev)); Hn.append(torch.empty(MB, N, device=dev, dtype=dt)); Ht.append(torch.zeros(N, MB, device=dev, dtype=dt))
            bs.append(rnd(N))
            # as the step launches it: bf16 runs write layers 0, 1 in bf16 only (both orientations), the last layer in fp32 only
            only_elem = a.bf16 and l < 2
            args.append(NtArgs(Xs[net].data_ptr(), Kp, Ws[net].data_ptr(), Kp, MB, N, Kp, Kp, None if only_elem else Hs[net].data_ptr(), N, 0,
                               Hn[net].data_ptr() if only_elem else None, N, Ht[net].data_ptr() if l < 2 else None, MB, bs[net].data_ptr(), None, 0, None, 0, None))
        fl = sum(2.0 * MB * N * ins[l][net] for net in range(3))

        def chk_f():
            ref = torch.nn.functional.elu(Xs[2].float() @ Ws[2].float().t() + bs[2])
            got = Hn[2].float() if (a.bf16 and l < 2) else Hs[2]
            e1_ = float((got - ref).abs().max() / ref.abs().max())
            e2_ = float((Ht[2].float().t() - ref).abs().max() / ref.abs().max()) if l < 2 else 0.0
            return max(e1_, e2_)
        total_us += run("forward L%d  [%d x %d x (%d|%d|%d)] x 3 nets" % (l, MB, N, *ins[l]), EPI_FWD, args, fl, check=chk_f); total_fl += fl
        if a.ablate and l == 1:
            for tag, keep in (("no output copy (reduction loop alone)", ()), ("[i][j] copy only", ("Cf", "Cn")), ("transposed copy only", ("Ct",))):
                ab = [NtArgs.from_buffer_copy(bytes(x)) for x in args]
                for x in ab:
                    for f in ("Cf", "Cn", "Ct"):
                        if f not in keep:
                            setattr(x, f, None)
                run("  ablation, forward L1: " + tag, EPI_FWD, ab, fl)
        # ---- weight gradient: G = dY^T X, split over the rows
        # splits as sdxpk_big_step picks them per layer: about a round and a half of tiles on the 512 workgroup slots, >= 4 chunks per split
        tiles = 3 * ((N + 127) // 128) * ((max(ins[l]) + 127) // 128)
        S = max(1, min((768 + tiles - 1) // tiles, max(1, MB // (4 * KC)), min(16, max(1, (MB + 511) // 512))))
        kc = ((MB + S - 1) // S + KC - 1) // KC * KC
        dYt, Xt, parts, args = [], [], [], []
        for net in range(3):
            K = ins[l][net]
            dYt.append((rnd(N, MB) * 0.1).to(dt).contiguous()); Xt.append(rnd(K, MB).to(dt).contiguous())
            pz = N * K + N
            parts.append(torch.empty(S, pz, device=dev))
            args.append(NtArgs(dYt[net].data_ptr(), MB, Xt[net].data_ptr(), MB, N, K, MB, kc, parts[net].data_ptr(), K, pz, None, 0, None, 0, None, None, 0, None, 0,
                               parts[net].data_ptr() + 4 * N * K))
        fl = sum(2.0 * MB * N * ins[l][net] for net in range(3))

        def chk_w():
            K = ins[l][2]
            tot = parts[2].sum(0)
            ref = dYt[2].float() @ Xt[2].float().t()
            e1_ = float((tot[:N * K].reshape(N, K) - ref).abs().max() / ref.abs().max())
            rs = dYt[2].float().sum(1)
            e2_ = float((tot[N * K:] - rs).abs().max() / rs.abs().max())
            return max(e1_, e2_)
        total_us += run("weight grad L%d [%d x (%d|%d|%d) x %d] x 3, %d splits" % (l, N, *ins[l], MB, S), EPI_TN, args, fl, splits=S, check=chk_w); total_fl += fl
        # ---- data gradient: dX = (dY W) * ELU'(H_prev)
        if l > 0:
            Kl = units[l - 1]
            dY, Wt, Hp, Hpt, dXn, dXt, args = [], [], [], [], [], [], []
            for net in range(3):
                dY.append((rnd(MB, N) * 0.1).to(dt).contiguous()); Wt.append((rnd(Kl, N) / N ** 0.5).to(dt).contiguous()); Hp.append(rnd(MB, Kl).to(dt).contiguous()); Hpt.append(Hp[-1].t().contiguous())
                dXn.append(torch.empty(MB, Kl, device=dev, dtype=dt)); dXt.append(torch.zeros(Kl, MB, device=dev, dtype=dt))
                args.append(NtArgs(dY[net].data_ptr(), N, Wt[net].data_ptr(), N, MB, Kl, N, N, None, 0, 0, dXn[net].data_ptr() if l > 1 else None, Kl,
                                   dXt[net].data_ptr(), MB, None, Hp[net].data_ptr(), Kl, Hpt[net].data_ptr(), MB, None))
            fl = 3 * 2.0 * MB * N * Kl

            def chk_d():
                hp = Hp[2].float()
                ref = (dY[2].float() @ Wt[2].float().t()) * torch.where(hp > 0, torch.ones_like(hp), hp + 1)
                e = float((dXt[2].float().t() - ref).abs().max() / ref.abs().max())
                return max(e, float((dXn[2].float() - ref).abs().max() / ref.abs().max())) if l > 1 else e
            total_us += run("data grad L%d->L%d [%d x %d x %d] x 3 nets" % (l, l - 1, MB, Kl, N), EPI_NN, args, fl, check=chk_d); total_fl += fl
            if a.ablate and l == 1:
                ab = [NtArgs.from_buffer_copy(bytes(x)) for x in args]
                for x in ab:
                    x.Cn = None; x.Ct = None
                run("  ablation, data grad L1->L0: no output copy (reduction loop alone)", EPI_NN, ab, fl)
    ln = "all eight trunk products of one optimiser step: %.1f us, %.1f TFLOP/s = %.1f %% of the %s dense peak (%.0f)" % (
        total_us, total_fl / total_us / 1e6, 100 * total_fl / total_us / 1e6 / peak, "bf16" if a.bf16 else "fp32", peak)
    lines.append(ln)
    print(ln)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write("# python tools/time_gemm_nt.py --mb %d%s   (k_gemm_nt of csrc/sdx_gemm_nt.h; HIP events over %d launches each)\n" % (MB, " --bf16" if a.bf16 else "", a.reps))
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
