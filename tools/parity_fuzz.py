#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random image sizes, sample grids, bandwidths, iteration counts and K against
the literal fp64 oracle, every formulation.  Prints one line per failing case and a summary; exit code 1 on failure.
    python tools/parity_fuzz.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    big = len(sys.argv) > 3 and sys.argv[3] == "big"   # sample grids up to 32 x 36 (chunked pair histogram, p > 256)
    rng = np.random.default_rng(seed)
    nle, oracle = entry.load_package(), entry.load_oracle()
    ctx = nle.Context(0)
    MODES = (0, 2, 4, 5, 1, 3)  # auto, tables (fp64), literal decomposition in fp64, streamed fp64; the two opt-in fp32 forms
    FP64 = (0, 2, 4, 5)
    worst = {m: 0.0 for m in MODES}
    skipped = {m: 0 for m in MODES}
    bad = {m: 0 for m in MODES}
    borderline = {"cases": 0, "rank_mismatch": 0, "worst": 0.0}
    done = 0
    while done < n_cases:
        if big:
            H, W = int(rng.integers(70, 200)), int(rng.integers(80, 240))
            nr, nc = int(rng.integers(8, 33)), int(rng.integers(10, 37))
            if nr > H // 2 or nc > W // 2:
                continue
        else:
            H, W = int(rng.integers(24, 140)), int(rng.integers(24, 160))
            nr, nc = int(rng.integers(2, 11)), int(rng.integers(2, 13))
            if nr > H // 3 or nc > W // 3:
                continue
        hx = float(rng.choice([8.0, 20.0, 60.0, 200.0, 1e4])) * float(rng.uniform(0.7, 1.3))
        hy = float(rng.choice([10.0, 30.0, 80.0])) * float(rng.uniform(0.7, 1.3))
        T, K, L = int(rng.integers(1, 16)), int(rng.integers(1, 40)), int(rng.integers(1, 6))
        kind = rng.integers(0, 3)
        if kind == 0:
            x = oracle.synthetic_luminance(H, W, seed=int(rng.integers(1, 1 << 30)))
        elif kind == 1:   # few grey levels + noise: flat regions, near-duplicate samples
            x = np.clip(rng.choice([40.0, 90.0, 150.0, 220.0], size=(H, W)) + rng.integers(-3, 4, (H, W)), 0, 255)
        else:             # smooth ramp + noise
            rr, cc = np.mgrid[0:H, 0:W]
            x = np.clip(np.rint(30 + 180 * (rr / H) * (cc / W) + rng.integers(-20, 21, (H, W))), 0, 255)
        try:
            V_o, S_o, inter = oracle.train_filter(x, nr, nc, hx, hy, T, K, return_intermediates=True)
        except Exception as e:  # noqa: BLE001  (degenerate case for the oracle itself)
            continue
        lam = inter["lam"]
        w_all = np.linalg.eigvalsh(inter["Ka"])[::-1]
        # inputs whose rank cut is borderline (an eigenvalue of Ka within a factor 100 of the 1e-10 cut on either side):
        # a separate, reported class -- the README images live there (tests/test_readme_pairs_gpu.py) -- on which the
        # fp64 formulations must still keep the oracle's rank and meet the bar whenever the oracle itself is stable
        r = lam.size
        is_borderline = (r < w_all.size and abs(w_all[r]) > 1e-12) or lam[-1] < 1e-8
        p = inter["Ka"].shape[0]
        Y_o = oracle.apply_layers(V_o, S_o, x, L).reshape(L, -1)
        # How sensitive is the reference algorithm itself on this input?  The same oracle pipeline with independent
        # relative noise of 1e-12 on every affinity (what any re-implementation's rounding looks like, scaled up):
        # amplification = per-layer output change / 1e-12.  Inputs on which Sinkhorn leaves W far from balanced (top
        # eigenvalue of Q >> 1, indefinite W_A), near-singular K_A, or a detail layer that is a tiny difference
        # 1 - lambda amplify by 1e6 ... 1e12; no two implementations agree there, so the bar is only asserted where
        # noise of the formulation's own arithmetic (1e-15 fp64, 1e-7 fp32 affinities) stays below 2e-5.
        perm, Ka0, Kab0 = oracle.compute_kernel(x, nr, nc, hx, hy)
        prng = np.random.default_rng(12345)
        Ka1 = Ka0 * (1.0 + 1e-12 * prng.standard_normal(Ka0.shape))
        Ka1 = np.tril(Ka1) + np.tril(Ka1, -1).T
        Kab1 = Kab0 * (1.0 + 1e-12 * prng.standard_normal(Kab0.shape))
        try:
            ev1, phi1 = oracle.nystrom_approximation(Ka1, Kab1)
            Wa1, Wab1, _, _ = oracle.sinkhorn_with_scalings(phi1, ev1, T)
            Vp, S_p = oracle.orthogonalize(Wa1, Wab1, K)
        except Exception:  # noqa: BLE001
            continue
        if S_p.size != S_o.size:
            continue
        if is_borderline:
            inter_p = oracle.eigen_decomposition(Ka1)[1]
            if inter_p.size != r:
                continue          # the oracle's own rank flips under 1e-12 noise: no implementation is defined here
        V_p = np.empty_like(Vp)
        V_p[perm] = Vp
        Y_p = oracle.apply_layers(V_p, S_p, x, L).reshape(L, -1)
        amp = max(rel(Y_p[j], Y_o[j]) for j in range(L)) / 1e-12
        # Second probe: the SAME literal oracle with LAPACK's QR-iteration eigensolver (dsyev) in place of numpy's
        # divide-and-conquer one (dsyevd) for its three eigensolves.  Both are backward stable; where they disagree
        # (eigenpairs of Q at the 1e-10 noise floor kept among the K: their 1 - lambda weight is ~1 in layer 0 only)
        # the reference's output is decided by its eigensolver's rounding and no other implementation can match it.
        import scipy.linalg as sl
        orig_eigh = np.linalg.eigh
        np.linalg.eigh = lambda M, UPLO="L": sl.eigh(M, lower=(UPLO == "L"), driver="ev")
        try:
            V_q, S_q = oracle.train_filter(x, nr, nc, hx, hy, T, K)
            Y_q = oracle.apply_layers(V_q, S_q, x, L).reshape(L, -1)
            solver_sens = max(rel(Y_q[j], Y_o[j]) for j in range(L)) if S_q.size == S_o.size else float("inf")
        except Exception:  # noqa: BLE001
            solver_sens = float("inf")
        finally:
            np.linalg.eigh = orig_eigh
        done += 1
        if is_borderline:
            borderline["cases"] += 1
        for mode in (tuple(m for m in MODES if m != 3) if (big and p > 256) else MODES):   # 0 = auto: what a caller gets
            ctx.set_mode(mode)
            try:
                f = nle.NLEFilter(ctx).train_filter(x.astype(np.float32), nr, nc, hx, hy, T, K)
                Y = f.apply_layers(x.astype(np.float32), L).cpu().numpy().astype(np.float64)
                info = f.info()
                errs = [rel(Y[j], Y_o[j]) for j in range(L)]
                ev = rel(f.eigvals, S_o) if info["K"] == S_o.size else float("inf")
                f.close()
            except Exception as e:  # noqa: BLE001
                if mode != 0 and "supports at most" in repr(e):
                    # a FORCED formulation refusing a case outside its limits (e.g. 36 sample rows on the table form:
                    # auto mode takes the fp64 decomposition there) is not a parity failure
                    skipped[mode] += 1
                    ctx.set_mode(0)
                    continue
                errs, ev = [float("inf")], float("inf")
                print("EXC", mode, (H, W, nr, nc, hx, hy, T, K, L), repr(e)[:200])
            finally:
                ctx.set_mode(0)
            m = max(errs)
            # noise each formulation injects into the affinities: fp32 (1e-7) for the materialised and exp forms,
            # fp64 (1e-15) for the tables; predicted output error = noise x amplification
            took_tables = mode in FP64     # integer planes and grids <= 32 x 36 here: auto takes the table form
            # fp64 forms: rounding acts like relative noise of 1e-15 on the affinities (the factored Sinkhorn update
            # keeps it there also when Ka is near singular: tools/readme_pair_sensitivity.py)
            noise = 1e-15 if took_tables else 1e-7
            predicted = max(noise * amp, solver_sens)
            if predicted > 2e-5:
                skipped[mode] += 1
                continue   # not well posed for this formulation's arithmetic
            tol = 1e-4
            worst[mode] = max(worst[mode], m)
            if is_borderline and mode in FP64:
                if info["r"] != r:
                    borderline["rank_mismatch"] += 1
                borderline["worst"] = max(borderline["worst"], m)
            if not (m < tol and ev < 1e-3):
                bad[mode] += 1
                out_dir = os.path.join(ROOT, "gpurun_out")
                if os.path.isdir(out_dir):   # keep the input for a replay
                    np.savez(os.path.join(out_dir, f"fuzz_fail_{seed}_{done}_{mode}.npz"), x=x,
                             params=np.array([nr, nc, hx, hy, T, K, L, mode], dtype=np.float64))
                print("FAIL mode", mode, (H, W, nr, nc, round(hx, 2), round(hy, 2), T, K, L), "kind", int(kind), "p", p, "r", r,
                      "lam_min %.2e" % lam[-1], "amp %.1e" % amp, "solver_sens %.1e" % solver_sens, "layers", ["%.1e" % e for e in errs], "eig %.1e" % ev, flush=True)
    names = {0: "auto", 2: "tables_f64", 4: "materialised_f64", 5: "streamed_f64", 1: "materialised_f32 (opt-in)", 3: "phi_free_exp_f32 (opt-in)"}
    print(f"{done} cases ({borderline['cases']} with a borderline rank cut); per formulation: asserted / failures / worst per-layer error")
    for m in MODES:
        print(f"  {names[m]:28s} {done - skipped[m]:4d} / {bad[m]:3d} / {worst[m]:.2e}")
    print(f"borderline class, fp64 formulations: rank mismatches {borderline['rank_mismatch']}, worst per-layer error {borderline['worst']:.2e}")
    return 1 if any(bad[m] for m in FP64) else 0


if __name__ == "__main__":
    sys.exit(main())
