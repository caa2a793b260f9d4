#!/usr/bin/env python3
"""Headline benchmark of the halo-model hot path on MI355X (BASELINE.json / SURVEY §8d).

    python bench.py --gpus N --steps K --warmup W

Workload (configs[2], "Config 3"): zs=linspace(0.01,3,32), ms=geomspace(2e10,1e17,512),
ks=geomspace(1e-4,100,4096); analytic NFW + Battaglia AGN electron profile (nxs=5000,
xmax=20) + HOD 'g' (mthresh=10^10.5); all six auto/cross spectra, 1-halo + 2-halo.
One STEP = one full pass of the path: sigma^2 -> n(z,m), b(z,m) -> c, rvir -> NFW u(k) ->
mass conversion -> Battaglia rows -> integrand/FFT/interpolation -> HOD -> fused mass
integrals of the six spectra (-> RCCL all-gather of the z-slabs when N>1).  Inputs (grids,
P(k)) are resident in HBM before the timed region; results stay in HBM.  The launches of a
step are captured once as a HIP graph and replayed (--no-graph issues them one by one).

N>1: one process per GPU.  Started as the driver starts it for N=1 (`python bench.py --gpus N`,
no WORLD_SIZE in the environment) this script spawns the N ranks itself - fresh child
processes from a parent that never touches the GPU - and relays rank 0's JSON line; under
`torch.distributed.run` (RANK/LOCAL_RANK/WORLD_SIZE set) it is one of the ranks.  The SAME
grid is partitioned in contiguous z-slabs (strong scaling) and gathered with one RCCL group
call per step.  The host side uses no torch: rendezvous is a file, barriers/gathers are RCCL
calls inside libhmgrid.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import uuid

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"),
         ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VALU_CYCLES = 4      # a wave64 fp64 VALU instruction occupies its SIMD-32 for 4 cycles
N_SIMD, CLOCK_HZ = 1024, 2.4e9
PROFILE_ROUND = "r06"
PROFILE_DIR = os.path.join(REPO, "profiles", PROFILE_ROUND)
BRACKET_EVERY = int(os.environ.get("HMG_BENCH_BRACKET_EVERY", "8"))   # kernel-level HIP events ride on every 8th timed step


def workload(nz=32, nm=512, nk=4096):
    zs = np.linspace(0.01, 3.0, nz)
    ms = np.geomspace(2e10, 1e17, nm)
    ks = np.geomspace(1e-4, 100, nk)
    return zs, ms, ks


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no rank environment
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv, cmd=None):
    """Start one fresh process per GPU with RANK/LOCAL_RANK/WORLD_SIZE set, relay rank 0's stdout.
    The parent makes no GPU call (it does not even import the package), so nothing is re-executed
    from a process that has initialised the device.  The ranks are SUPERVISED: all of them are polled,
    the first non-zero exit terminates the others (a rank that dies before the communicator exists would
    otherwise leave its peers waiting in ncclCommInitRank or at the rendezvous file), an overall deadline
    bounds the launch, and the failing rank's stderr tail is reported.  Returns the exit code."""
    import socket
    import tempfile
    import threading
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if cmd is None:            # (tests hand in another rank program to exercise the supervision)
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-run"]
    deadline_s = float(os.environ.get("HMG_LAUNCH_DEADLINE", "540"))      # under the driver's 600 s
    envs = []
    launch_nonce = uuid.uuid4().hex      # one per launch: tells this launch's rendezvous file from a leftover of the same name
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HMG_LAUNCH_TAG=f"{port}_{os.getpid()}",
                   HMG_LAUNCH_NONCE=launch_nonce)
        # dmabuf IPC is the only mode the host driver of this pool supports; RCCL's peer set-up needs it
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a supervised rank never needs to wait for the id longer than the slowest first library page-in
        env.setdefault("HMG_RDZV_TIMEOUT", "240")
        envs.append(env)
    if args.dry_run:
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HMG_LAUNCH_TAG", "HMG_LAUNCH_NONCE",
                "HSA_ENABLE_IPC_MODE_LEGACY", "HMG_RDZV_TIMEOUT")
        print(json.dumps({"dry_run": True, "n_ranks": n, "cmd": cmd, "deadline_s": deadline_s,
                          "rank_env": [{k: e[k] for k in keys} for e in envs]}))
        return 0
    errs = [tempfile.TemporaryFile() for _ in range(n)]
    procs = [subprocess.Popen(cmd, env=envs[r], stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                              stderr=errs[r]) for r in range(n)]
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_end = time.monotonic() + deadline_s
    failed, why = None, ""
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed, why = bad[0], f"rank {bad[0]} exited with code {rcs[bad[0]]}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > t_end:
            failed, why = -1, f"launch exceeded its deadline of {deadline_s:.0f} s"
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    sys.stdout.write((out0[0] if out0 else b"").decode())
    sys.stdout.flush()
    if failed is not None:
        rcs = [p.returncode for p in procs]
        sys.stderr.write(f"bench.py launcher: {why}; rank exit codes {rcs}\n")
        show = failed if failed >= 0 else 0
        errs[show].seek(0)
        tail = errs[show].read().decode(errors="replace").strip().splitlines()[-15:]
        sys.stderr.write(f"---- stderr tail of rank {show} ----\n" + "\n".join(tail) + "\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------
def _cpu_inputs(zs, ks, p):
    import hmvec_amd as hm
    from oracle import hmref
    cos = hm.Cosmology(dict(p), accuracy="low", engine="analytic")
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    return hmref.CosmoInputs(h=cos.h, omm0=cos.omm0, ombh2=p["ombh2"],
                             rho_crit_0=float(cos.rho_critical_z(0.0)), rho_crit_zs=cos.rho_critical_z(zs),
                             Pzk=cos.P_lin_approx(ks, zs), sPzk=cos.P_lin_approx(ksig, zs), ks_sigma2=ksig,
                             h_of_z_zs=cos.h_of_z(zs))


def _cpu_pass(ci, z, ms, ks, p, nxs, xmax=20.0):
    """One pass of the oracle over the redshifts z: per-stage seconds and the six spectra."""
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    t = [time.perf_counter()]
    o = hmref.RefHaloModel(ci, z, ks, ms, p, skip_nfw=True)
    t.append(time.perf_counter())
    o.add_nfw_profile("nfw")
    t.append(time.perf_counter())
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], nxs, xmax)
    t.append(time.perf_counter())
    o.add_hod("g", mthresh=10 ** 10.5 + z * 0.0)
    t.append(time.perf_counter())
    out = [o.get_power(a, b) for a, b in PAIRS]
    t.append(time.perf_counter())
    return np.diff(t), out


def _cpu_worker(job):
    """All-core variant: one process per z-slab, each running the single-threaded oracle."""
    zs, ms, ks, nxs, reps = job[:5]
    xmax = job[5] if len(job) > 5 else 20.0
    from hmvec_amd.params import default_params
    p = dict(default_params)
    ci = _cpu_inputs(zs, ks, p)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        _cpu_pass(ci, zs, ms, ks, p, nxs, xmax)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def cpu_baseline(zs, ms, ks, nz_sample, nxs, allcore=True, xmax=20.0):
    """The CPU oracle (numpy restatement of the reference, pinned by tests/golden) timed on this
    box's host cores, BEFORE the process touches the GPU: 1 warm-up + best of 3, per stage and
    total, on a z-subsample of the same workload (numpy elementwise ops, trapz, pocketfft and
    interp are single-threaded: 1 core effective); plus an all-core variant - the full z grid cut
    into slabs over a process pool - so that the baseline beside the GPU number is not one core
    of many.  Returns (record, sample indices, spectra of the sample for the parity check)."""
    import multiprocessing as mp
    import platform
    import scipy
    from hmvec_amd.params import default_params
    p = dict(default_params)
    sel = np.linspace(0, zs.size - 1, nz_sample).round().astype(int)
    z = zs[sel]
    ci = _cpu_inputs(z, ks, p)
    stages = ["mass_function", "nfw", "battaglia_fft", "hod", "six_spectra"]
    runs = []
    for _ in range(4):                      # first one is the warm-up
        st, out = _cpu_pass(ci, z, ms, ks, p, nxs, xmax)
        runs.append(st)
    runs = np.array(runs[1:])
    best = runs[np.argmin(runs.sum(axis=1))]
    pts = len(PAIRS) * z.size * ms.size * ks.size
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: d.get(k) for k in ("internal_api", "num_threads")} for d in threadpool_info()]
    except Exception:
        pools = None
    avail = len(os.sched_getaffinity(0))
    rec = dict(value=pts / best.sum(), unit="grid-points/s", cores=1, kind="port",
               sample=f"{z.size} of {zs.size} redshifts x {ms.size} x {ks.size}, nxs={nxs}, xmax={xmax:g}, 6 spectra; 1 warm-up + "
                      f"best of 3 passes ({best.sum():.2f} s best, {runs.sum(axis=1).max():.2f} s worst), numpy single-thread",
               stages_s=dict(zip(stages, best.tolist())), seconds=float(best.sum()),
               full_grid_seconds_extrapolated=float(best.sum() * zs.size / z.size),
               host={"cpu_model": model, "cpu_count": os.cpu_count(), "affinity": avail,
                     "numpy": np.__version__, "scipy": scipy.__version__, "python": platform.python_version(),
                     "threadpools": pools})
    if allcore and avail > 1:
        nproc = 1
        for cand in (zs.size, 16, 8, 4, 2):   # a divisor of nz the box can run at once: one redshift per process (32) where it can
            if cand <= avail and zs.size % cand == 0:
                nproc = cand
                break
        per = zs.size // nproc
        jobs = [(zs[i * per:(i + 1) * per], ms, ks, nxs, 2, xmax) for i in range(nproc)]
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(nproc) as pool:      # forked before any GPU initialisation
            bests = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
        slowest = max(bests)
        rec["all_cores"] = dict(value=len(PAIRS) * zs.size * ms.size * ks.size / slowest, unit="grid-points/s",
                                cores=nproc, pool_size=nproc, hardware_threads=avail, seconds=slowest,
                                sample=f"all {zs.size} redshifts as {nproc} z-slabs of {per} over a process pool "
                                       f"(one single-threaded oracle per core; slowest slab's best of 2; pool wall {wall:.1f} s)")
    return rec, sel, out


# ------------------------------------------------------------------------------------------------
# BASELINE configs[0]/[1]: the README usage sequence on 20 x 200 x 1001 (README.rst:52-90)
# ------------------------------------------------------------------------------------------------
def readme_config2(ctx, with_cpu=True):
    """Facade wall time (host work included) of the README sequence - constructor, add_battaglia_profile(nxs=5000),
    add_hod, twelve get_power_1halo/_2halo calls returning host arrays - on the reference's own grid, its parity
    against the strided sample of the UNMODIFIED reference's run (tests/golden/readme_c1.npz) and the oracle's time
    for the same sequence on this box's host."""
    import hmvec_amd as hm
    from hmvec_amd.params import battaglia_defaults, default_params
    zs = np.linspace(0., 3., 20); ms = np.geomspace(2e10, 1e17, 200); ks = np.geomspace(1e-4, 100, 1001)
    names = ["nfw", "electron", "g"]
    pairs = [(a, b) for i, a in enumerate(names) for b in names[i:]]

    def seq():
        t0 = time.perf_counter()
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
        t1 = time.perf_counter()
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.)
        out = {}
        for a, b in pairs:
            out[(a, b)] = (h.get_power_1halo(a, b), h.get_power_2halo(a, b))
        t2 = time.perf_counter()
        return h, out, (t1 - t0, t2 - t1)

    seq(); seq()                                  # first launches, pinned staging, FFT tables
    runs = [seq() for _ in range(7)]
    h, out, _ = runs[-1]
    tc = sorted(r[2][0] for r in runs)[len(runs) // 2]
    tr = sorted(r[2][1] for r in runs)[len(runs) // 2]
    # COLD constructor: the warm figure above rides on the provider caches (T(k), the P(k',z) product
    # on the shared sigma^2 grid, its device layout) that a repeated parameter set hits.  A sampler changes the cosmology
    # every step: fresh grid ARRAYS and a changed omch2 miss every one of them.  Median of 7, each with its own omch2.
    cold = []
    for i in range(7):
        zs_c, ks_c, ms_c = zs.copy(), ks.copy(), ms.copy()
        t0 = time.perf_counter()
        hc = hm.HaloModel(zs_c, ks_c, ms=ms_c, params={"omch2": 0.1198 * (1.0 + 1e-3 * (i + 1))}, accuracy="low",
                          engine="analytic", ctx=ctx)
        hc._ctx().sync()
        cold.append(time.perf_counter() - t0)
        del hc
    tcold = sorted(cold)[len(cold) // 2]
    rec = {"grid": "zs=20 (0..3) ms=200 ks=1001, analytic NFW + Battaglia AGN nxs=5000 xmax=20 + HOD(mthresh=10^10.5)",
           "ctor_ms": tc * 1e3, "ctor_ms_cold": tcold * 1e3,
           "ctor_note": "ctor_ms: the same parameters and grid objects again (provider caches hit: T(k), P(k',z) on the shared "
                        "sigma^2 grid and its device layout); ctor_ms_cold: fresh grid arrays and another omch2 per "
                        "construction (every cache misses: Eisenstein-Hu on the 10^4-point sigma^2 grid, upload, layout), "
                        "device work waited for",
           "profiles_hod_12_get_power_ms": tr * 1e3, "sequence_ms": (tc + tr) * 1e3,
           "sequence_ms_cold_ctor": (tcold + tr) * 1e3,
           "timing": "median of 7 sequences after 2 warm-up sequences; host wall incl. the provider's numpy, uploads "
                     "and the twelve (nz,nk) results copied to the host",
           "grid_points_per_s": len(pairs) * zs.size * ms.size * ks.size / (tc + tr)}
    gpath = os.path.join(REPO, "tests", "golden", "readme_c1.npz")
    if os.path.exists(gpath):
        with np.load(gpath, allow_pickle=False) as g:
            ki = slice(None, None, int(g["kstride"]))
            worst = 0.0
            for (a, b), (p1, p2) in out.items():
                for got, key in ((p1, f"P1h_{a}_{b}"), (p2, f"P2h_{a}_{b}")):
                    R = g[key]
                    tol = 1e-8 * np.abs(R) + 1e-12 * np.max(np.abs(R), axis=-1, keepdims=True)
                    worst = max(worst, float(np.max(np.abs(got[:, ki] - R) / tol)))
        rec["parity_worst_dP_over_tol"] = worst
        rec["parity_against"] = "tests/golden/readme_c1.npz: the unmodified reference's run of this sequence, every 20th k"
    if with_cpu:
        from oracle import hmref
        p = dict(default_params)
        ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
        t0 = time.perf_counter()
        ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                               rho_crit_zs=h.rho_critical_z(zs), Pzk=h.Pzk, sPzk=h.sPzk, ks_sigma2=ksig,
                               h_of_z_zs=h.h_of_z(zs))
        o = hmref.RefHaloModel(ci, zs, ks, ms, p)
        t1 = time.perf_counter()
        o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 5000, 20)
        o.add_hod("g", mthresh=10 ** 10.5 + zs * 0.)
        for a, b in pairs:
            o.get_power_1halo(a, b); o.get_power_2halo(a, b)
        t2 = time.perf_counter()
        rec["cpu_oracle"] = {"ctor_ms": (t1 - t0) * 1e3, "rest_ms": (t2 - t1) * 1e3, "sequence_ms": (t2 - t0) * 1e3,
                             "cores": 1, "kind": "port", "sample": "the whole sequence, one run"}
    return rec


# ------------------------------------------------------------------------------------------------
# the radial grid the reference's own callers use (examples/lensing_baryons.py:27, bin/tests.py:308)
# ------------------------------------------------------------------------------------------------
def stage_stream(c2, stage, n=12):
    """Milliseconds per call of a launch-only stage issued n times back to back between two events - no host
    synchronisation in between, i.e. at the clock and cache state of a stream of passes (the eager, host-synchronised
    stage times beside it start every launch from an idle GPU: 1.85 instead of 2.2 GHz measured inside the kernel)."""
    for _ in range(3):
        stage()
    c2.sync()
    c2.record(46)
    for _ in range(n):
        stage()
    c2.record(47)
    c2.sync()
    return c2.elapsed_ms(46, 47) / n


def long_grid_block(ctx, zs, ms, ks, mthr, pairs, reps=12):
    """add_battaglia_profile(xmax=50, nxs=30000) on the bench grid: milliseconds of the profile stage (HIP events)
    through the pruned long-grid route (+ chirp route for rows that need few modes) and through the chunked rocFFT
    route it replaces, the agreement of the two tensors, and the wall time of whole passes with that profile."""
    import hmvec_amd as hm
    from hmvec_amd import _native as nat
    out = {"profile": "add_battaglia_profile('electron', family='AGN', xmax=50, nxs=30000)",
           "grid": f"zs={zs.size} ms={ms.size} ks={ks.size}"}
    tens = {}
    for route, env in (("pruned", {}), ("pruned_no_chirp", {"HMG_CHIRP": "0"}), ("rocfft", {"HMG_PRUNED_FFT": "0"})):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        try:
            c2 = nat.Context(ctx.device)          # the route switches are read when a context is created
            h2 = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=c2)
            n = reps if route != "rocfft" else 3
            o1 = [c2.empty((zs.size, ks.size)) for _ in pairs]      # result buffers of every pass (a captured pass allocates nothing)
            o2 = [c2.empty((zs.size, ks.size)) for _ in pairs]

            def one(timed):
                h2.init_mass_function(ms)
                h2.add_nfw_profile("nfw", ignore_existing=True)
                if timed:
                    c2.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, 44, 45)
                h2.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000, ignore_existing=True)
                h2.add_hod("g", mthresh=mthr, ignore_existing=True)
                return h2.power_device_batch(pairs, outs1=o1, outs2=o2)
            one(False); one(False)
            c2.sync()
            t_fft = []
            t0 = time.perf_counter()
            for _ in range(n):
                one(True)
                c2.sync()
                t_fft.append(c2.elapsed_ms(44, 45))
            wall = (time.perf_counter() - t0) / n * 1e3
            out[route] = {"profile_stage_ms": float(np.median(t_fft)), "pass_wall_ms_eager": wall}
            if route == "pruned":
                # the same pass as ONE captured step, replayed: the number that compares with the headline ms_per_step
                gid = c2.capture(lambda: one(False))
                for _ in range(64):               # (the same clock / cache preconditioning as the headline loop)
                    c2.replay(gid)
                c2.sync()
                K = 40
                c2.record(46)
                for _ in range(K):
                    c2.replay(gid)
                c2.record(47)
                c2.sync()
                out["ms_per_step"] = c2.elapsed_ms(46, 47) / K
                out["ms_per_step_note"] = (f"HIP-graph replay of the whole pass with this profile, {K} replays between two events "
                                           "(the headline step with nxs = 5000 is timed the same way)")
                out["pruned"]["profile_stage_ms_stream"] = stage_stream(
                    c2, lambda: h2.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000, ignore_existing=True))
                # the numeric NFW branch at the reference's defaults (hmvec/params.py:59-60): nxs = 40000, xmax = 200
                t_ = []
                for i in range(2 + 5):
                    c2.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, 44, 45)
                    h2.add_nfw_profile("nfwnum", numeric=True, ignore_existing=True)
                    c2.sync()
                    if i >= 2:
                        t_.append(c2.elapsed_ms(44, 45))
                out["numeric_nfw"] = {"profile": "add_nfw_profile(numeric=True): nxs=40000, xmax=200 (hmvec/params.py:59-60)",
                                      "profile_stage_ms": float(np.median(t_)),
                                      "profile_stage_ms_stream": stage_stream(
                                          c2, lambda: h2.add_nfw_profile("nfwnum", numeric=True, ignore_existing=True))}
            if route != "pruned_no_chirp":
                tens[route] = h2.uk_profiles["electron"][::4, ::16]      # a strided sample of the (nz,nm,nk) tensor
            del h2
            c2.close()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
    out["max_abs_du_between_routes"] = float(np.max(np.abs(tens["pruned"] - tens["rocfft"])))
    # the third caller with a long grid: the tSZ notebook's pressure profile, whose support does not prune
    tsz = {"profile": "add_battaglia_pres_profile('y', family='pres', xmax=2, nxs=30000)  (examples/tSZ example.ipynb)"}
    pk = {}
    for route, env in (("narrow_band", {}), ("rocfft", {"HMG_BAND_FFT": "0"})):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        try:
            c2 = nat.Context(ctx.device)
            h2 = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=c2)
            t_ = []
            for i in range(2 + (reps if route != "rocfft" else 3)):
                c2.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, 44, 45)
                h2.add_battaglia_pres_profile("y", family="pres", xmax=2, nxs=30000, ignore_existing=True)
                c2.sync()
                if i >= 2:
                    t_.append(c2.elapsed_ms(44, 45))
            tsz[route] = {"profile_stage_ms": float(np.median(t_))}
            if route == "narrow_band":
                tsz[route]["profile_stage_ms_stream"] = stage_stream(
                    c2, lambda: h2.add_battaglia_pres_profile("y", family="pres", xmax=2, nxs=30000, ignore_existing=True))
            pk[route] = h2.pk_profiles["y"][::4, ::16]
            del h2
            c2.close()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
    scale = np.max(np.abs(pk["rocfft"]), axis=-1, keepdims=True)
    tsz["max_dp_over_rowmax_between_routes"] = float(np.max(np.abs(pk["narrow_band"] - pk["rocfft"]) / scale))
    out["tsz_pressure"] = tsz
    out["note"] = ("profile_stage_ms: HIP events around ONE launch after a host synchronisation (the GPU starts it from idle clocks); "
                   "profile_stage_ms_stream: the same stage issued 12 times back to back between two events, per call - the state a "
                   "stream of passes (and the graph replay behind ms_per_step) runs in.  pass_wall_ms_eager: one pass = mass function + NFW + this profile + HOD + the batched spectra, eager "
                   "launches, host-synchronised after every pass (the headline ms_per_step is a HIP-graph replay); "
                   "tolerance on u is 1e-12 absolute (tests/test_gpu_longgrid.py holds both routes to the reference's "
                   "own fixture case_f)")
    return out


# ------------------------------------------------------------------------------------------------
# byte and instruction models of the three large kernels
# ------------------------------------------------------------------------------------------------
def load_profile(name):
    path = os.path.join(PROFILE_DIR, name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


def power_bytes_moved(nzl, nm, nk, npair, nconst, vec):
    """HBM bytes one hmg_power_batch launch moves, from the launch's own skip rule: the NFW tensor is
    read whole; of the Battaglia tensor every (row, k-tile) whose tile lies inside the row's constant
    prefix (nconst[row] >= tile end) is not read.  Plus the coefficient rows, hints, P_lin and the
    2*npair output rows."""
    tile = 64 * vec
    ends = np.minimum(nk, (np.arange((nk + tile - 1) // tile) + 1) * tile)
    widths = np.diff(np.concatenate([[0], ends]))
    read = (nconst.reshape(-1, 1) < ends[None, :]) * widths[None, :]
    tens = 8.0 * nzl * nm * nk + 8.0 * float(read.sum())
    stride = 8                                    # coefficient doubles per (z,m): the compact rows of this batch's
                                                  # structure (hmgrid.hip pb_stride; the generic forms take 29)
    side = 8.0 * nzl * nm * (stride + 2)          # coefficient rows + hint count/value
    outs = 8.0 * nzl * nk * (2 * npair + 1)       # spectra + P_lin
    return tens + side + outs, tens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 0.6 ms each: clocks and caches settle after ~10
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nz", type=int, default=32)
    ap.add_argument("--nm", type=int, default=512)
    ap.add_argument("--nk", type=int, default=4096)
    ap.add_argument("--nxs", type=int, default=5000)
    ap.add_argument("--xmax", type=float, default=20.0,
                    help="extent of the radial grid of the Battaglia profile (the reference's own callers also use "
                         "xmax=50 with nxs=30000: examples/lensing_baryons.py:27, bin/tests.py:308)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue the launches of every step one by one")
    ap.add_argument("--graph", action="store_true",
                    help="time HIP-graph replays of the step (rounds 2-5's default) instead of eager launches from the "
                         "recorded list of native calls; the other mode is measured beside it either way")
    ap.add_argument("--lanes", action="store_true",
                    help="independent stages on separate HIP streams (concurrent graph branches)")
    ap.add_argument("--per-pair", action="store_true", help="six hmg_power launches instead of one hmg_power_batch")
    ap.add_argument("--no-long-grid", action="store_true",
                    help="skip the long-radial-grid block (nxs=30000, xmax=50 on the bench grid, both routes)")
    ap.add_argument("--no-readme", action="store_true",
                    help="skip the BASELINE configs[0]/[1] block (README sequence on 20 x 200 x 1001)")
    ap.add_argument("--no-limber", action="store_true",
                    help="skip the Config-5 leg (C_kk and C_kg at 2000 multipoles on the gathered spectra)")
    ap.add_argument("--stages", action="store_true",
                    help="also record per-stage HIP events (eager launches only; a few us of stream time each)")
    ap.add_argument("--cpu-sample-nz", type=int, default=8)
    ap.add_argument("--dry-run", action="store_true", help="launcher only: print the rank commands and exit")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_run):
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    args.gpus = world
    # Multi-process GPU work on this platform needs dmabuf IPC (the task environment exports it; keep it
    # if a launcher dropped it).  Set here, in the benchmark, before anything loads the HIP runtime -
    # the library itself leaves the process environment alone.
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    zs, ms, ks = workload(args.nz, args.nm, args.nk)
    K, W = args.steps, args.warmup

    # ---- CPU baseline first: nothing in this process has touched the GPU yet
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (the oracle materialises (nz_sample, nm, nxs) temporaries: fewer redshifts for the long radial grids)
        nz_cpu = min(args.cpu_sample_nz, zs.size) if args.nxs <= 10000 else min(2, zs.size)
        cpu = cpu_baseline(zs, ms, ks, nz_cpu, args.nxs, allcore=args.nxs <= 10000, xmax=args.xmax)

    if args.lanes:
        os.environ["HMG_LANES"] = "1"
    import hmvec_amd as hm
    from hmvec_amd import _native as nat
    from hmvec_amd.dist import RcclComm, ShardedSpectra, slab_bounds

    lo, hi = slab_bounds(zs.size, world, rank)
    zloc = zs[lo:hi]
    ctx = nat.Context(local_rank)
    tag = os.environ.get("HMG_LAUNCH_TAG") or \
        f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'solo')}"
    comm = RcclComm(ctx, rank, world, tag)
    rccl_rank, rccl_ranks = comm.info()          # (None, None) for a single rank: no communicator is created

    mthr = 10 ** 10.5 + zloc * 0.0
    h = hm.HaloModel(zloc, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    h.add_battaglia_profile("electron", family="AGN", xmax=args.xmax, nxs=args.nxs)
    h.add_hod("g", mthresh=mthr)
    spec = ShardedSpectra(h, comm, zs.size, PAIRS)
    npair = len(PAIRS)

    BR = {"power": nat.KERNEL_POWER, "nfw": nat.KERNEL_NFW, "fft": nat.KERNEL_PROFILE_FFT}
    EV_BR = {"power": (40, 41), "nfw": (42, 43), "fft": (44, 45)}      # bracket slots (HaloModel: 0-3, gather: 8-9)
    EV_STAGE = 48                                                       # 48..53 stage marks (--stages)

    def compute(brackets=False, stages=False):
        """The launches of one pass (everything but the gather): capturable."""
        def mark(i):
            if stages:
                ctx.record(EV_STAGE + i)
        def bracket(name):
            if brackets and not (name == "power" and args.per_pair):
                ctx.call("hmg_bracket_next", BR[name], EV_BR[name][0], EV_BR[name][1])
        mark(0)
        h.init_mass_function(ms)
        mark(1)
        bracket("nfw")
        h.add_nfw_profile("nfw", ignore_existing=True)
        mark(2)
        bracket("fft")
        h.add_battaglia_profile("electron", family="AGN", xmax=args.xmax, nxs=args.nxs, ignore_existing=True)
        mark(3)
        h.add_hod("g", mthresh=mthr, ignore_existing=True)
        mark(4)
        bracket("power")
        spec.launch_spectra(batched=not args.per_pair)
        mark(5)

    # ---- one eager pass sizes every scratch arena; then the step is captured (plain and bracketed)
    for _ in range(2):
        spec.wait_gathered(); compute(); spec.gather()
    ctx.sync()
    # Launch mode of the timed steps.  Default since round 6: the native calls of a pass recorded ONCE (Context.trace) and
    # re-issued eagerly per step without the Python facade (3 launches, ~0.02 ms of host time per step).  Rounds 2-5 replayed
    # a captured HIP graph of the same launches: on ROCm 7.2 consecutive graph launches leave ~4-9 us between the last
    # kernel of one replay and the first of the next (profiles/r06/slab4_timeline.txt), which a 0.1 ms thin-slab step
    # feels; eager launches on one stream run back to back.  --graph times the replay instead; the other mode is
    # measured beside the timed one either way (launch_modes in the JSON).
    args.call_list = not args.graph and not args.no_graph and not args.stages
    use_graph = args.graph and not args.no_graph and not args.stages
    can_both = not args.no_graph and not args.stages
    g_plain = g_brack = t_brack = t_plain = None
    if args.call_list:
        t_plain = ctx.trace(lambda: compute())
        t_brack = ctx.trace(lambda: compute(brackets=True))
        ctx.sync()
    events_in_graph = False
    if use_graph:
        g_plain = ctx.capture(lambda: compute())
        g_brack = ctx.capture(lambda: compute(brackets=True))
        ctx.replay(g_brack)
        ctx.sync()
        try:    # do events recorded by graph nodes carry timestamps on this runtime?
            t = ctx.elapsed_ms(*EV_BR["nfw"])
            events_in_graph = 0.0 < t < 100.0
        except nat.NativeError:
            events_in_graph = False
        if not events_in_graph:
            # They do not (ROCm 7.2): the bracketed steps are issued eagerly - as a recorded CALL LIST, not through
            # the Python facade (85 us of host work per pass, which made every bracketed step of a 0.1 ms thin-slab
            # pass host-bound and inflated ms_per_step by 10 %)
            t_brack = ctx.trace(lambda: compute(brackets=True))
            ctx.sync()

    kern_ms = {k: [] for k in BR}
    stage_ms = []
    t_wait = [0.0]        # host time spent waiting for bracket events (not launch work)

    def read_brackets():
        t_in = time.perf_counter()
        _read_brackets()
        t_wait[0] += time.perf_counter() - t_in

    brackets_missing = {}     # bracket name -> times its events were found unrecorded (reported in the JSON)

    def _read_brackets():
        for k in BR:
            if not (k == "power" and args.per_pair):
                try:
                    kern_ms[k].append(ctx.elapsed_ms(*EV_BR[k]))
                except nat.NativeError as e:
                    # Only ONE failure is expected here: a stage that rode in another stage's launch leaves its bracket
                    # unrecorded.  Anything else (a device fault surfaced by the event wait, a bad slot) is a real error.
                    if "never recorded" not in str(e):
                        raise
                    brackets_missing[k] = brackets_missing.get(k, 0) + 1
        if args.stages:
            stage_ms.append([ctx.elapsed_ms(EV_STAGE + j, EV_STAGE + j + 1) for j in range(5)])

    def step(timed_index=None):
        """One full pass.  Every BRACKET_EVERY-th timed step carries the kernel brackets; their event
        times are read just before the next bracketed step overwrites them (the GPU still has the
        steps in between queued, so the host-side wait opens no bubble)."""
        bracketed = timed_index is not None and timed_index % BRACKET_EVERY == 0
        if bracketed and timed_index > 0:
            read_brackets()
        spec.wait_gathered()
        if use_graph and (not bracketed or events_in_graph):
            ctx.replay(g_brack if bracketed else g_plain)
        elif t_brack is not None and (use_graph or args.call_list):
            ctx.run_trace(t_brack if bracketed else t_plain)
        else:
            compute(brackets=bracketed, stages=bracketed and args.stages)
        spec.gather()

    # Clock / cache preconditioning before the W warm-up steps: a sub-millisecond step leaves the GPU in
    # its idle power state for the first few milliseconds, so the steady state the metric is about needs
    # ~50 ms of work first.  Untimed, disclosed in the JSON.
    PRECONDITION = 64
    for _ in range(PRECONDITION + W):
        step()
    comm.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for s in range(K):
        step(s)
    t_issue = time.perf_counter() - t0 - t_wait[0]   # host time to enqueue K steps (launches are asynchronous)
    comm.barrier()
    ctx.sync()
    dt = time.perf_counter() - t0
    read_brackets()
    dt_all = comm.allgather_host([dt]).reshape(-1) if world > 1 else np.array([dt])
    dt_max = float(dt_all.max())
    # The collective of a step, separately: in the timed loop the gather runs on the communication
    # lane under the next pass, so its cost is hidden or not depending on the slab; here eight UNTIMED steps are issued
    # one at a time and the event pair around the gather (ready: after the mass integrals on the main lane, done: after
    # the RCCL group on the communication lane) is read after each - duration of the collective incl. the wait for the
    # slowest rank.
    gather_ms = None
    if world > 1 and spec._gather:
        g_ = []
        for _ in range(8):
            step()
            ctx.sync()
            g_.append(ctx.elapsed_ms(spec._EV_SPECTRA, spec._EV_GATHERED))
        g_all = comm.allgather_host([float(np.median(g_))]).reshape(-1)
        gather_ms = {"median_ms_this_rank": float(np.median(g_)), "max_over_ranks_ms": float(g_all.max()),
                     "min_over_ranks_ms": float(g_all.min()),
                     "note": "8 untimed steps after the timed loop, host-synchronised one by one; HIP events around the RCCL "
                             "group on the communication lane (ready -> done).  In the timed loop this collective overlaps "
                             "the next pass"}
    kms = {k: (float(np.mean(v)) if v else None) for k, v in kern_ms.items()}

    # ---- the other launch mode, beside the timed one (same kernels; 40 steps after 16 warm-up steps, max over ranks)
    other_mode = None
    if can_both:
        if args.call_list:
            g_other = ctx.capture(lambda: compute())
            issue = lambda: ctx.replay(g_other)                 # noqa: E731
            other_name = "hip-graph replay"
        else:
            t_other = ctx.trace(lambda: compute())
            issue = lambda: ctx.run_trace(t_other)              # noqa: E731
            other_name = "eager launches from a recorded call list"
        def other_step():
            spec.wait_gathered(); issue(); spec.gather()
        for _ in range(16):
            other_step()
        comm.barrier(); ctx.sync()
        t1 = time.perf_counter()
        for _ in range(40):
            other_step()
        comm.barrier(); ctx.sync()
        dt_o = time.perf_counter() - t1
        dt_o = float(comm.allgather_host([dt_o]).max()) if world > 1 else dt_o
        other_mode = {"mode": other_name, "ms_per_step": dt_o / 40 * 1e3, "steps": 40}
    g_count = g_plain if g_plain is not None else (g_other if (can_both and args.call_list) else None)

    nzl, nm_, nk_, nxs = zloc.size, ms.size, ks.size, args.nxs
    B = nzl * nm_
    tens_bytes = 8.0 * B * nk_

    if rank != 0:
        comm.barrier()
        comm.close()
        ctx.close()
        return

    # ---- bytes: the implementation's own model and, for the default configuration,
    # the PMC counters of profiles/<round> (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
    default_cfg = ((args.nz, args.nm, args.nk, args.nxs, args.xmax) == (32, 512, 4096, 5000, 20.0) and world == 1
                   and not args.per_pair)
    sha = nat.kernel_source_sha16()
    stale = []

    def stored(name):
        """A stored counter profile counts only for the build it was measured on."""
        d = load_profile(name) if default_cfg else None
        if d is not None and d.get("source_sha16") != sha:
            stale.append(f"{name}: measured on kernel sources {d.get('source_sha16')}, this build is {sha}")
            return None
        return d

    pmc = stored("pmc_traffic.json")
    sq = stored("sq_issue_counters.json")
    # what the bytes of a kernel entry are when no counter profile of THIS build is on file - said in the roofline block
    # itself, not only in stale_profiles_ignored
    stale_pmc = [t for t in stale if t.startswith("pmc_traffic.json")]
    model_source = ("model (stored profile stale: " + stale_pmc[0].split(": ", 1)[1] + ")") if stale_pmc else \
        ("model (DESIGN.md section 4; no stored counter profile applies to this configuration)")
    grouped = h._groups
    # Tensor group (round 6, hmg_group_tensors): the NFW rows ride in the profile rows' launch - no NFW launch, its bracket
    # is never recorded, and the "fft" bracket times the one kernel that writes both tensors.
    tensor = grouped and kms["nfw"] is None and kms["fft"] is not None
    KNAME = {"power": "power_batch_kernel", "nfw": "rows_group_kernel" if grouped else "nfw_kernel",
             "fft": ("tensor_group_kernel" if tensor else "profile_group_kernel") if grouped else "profile_fused_kernel"}

    def pmc_bytes(sub):
        if not pmc:
            return None
        for name, rec in pmc["kernels"].items():
            if sub in name:
                return rec["hbm_bytes_per_launch_corrected"]
        return None

    def valu(sub):
        if not sq:
            return None
        for name, rec in sq["kernels"].items():
            if sub in name and rec.get("SQ_WAVES"):
                per_wave = rec["SQ_INSTS_VALU"] / rec["SQ_WAVES"]
                return {"valu_insts_per_wave": per_wave,
                        "valu_issue_bound_ms": rec["SQ_INSTS_VALU"] * FP64_VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3}
        return None

    hint_n = h.uk_profiles.hint("electron")[0]
    vec = 2 if nk_ % 2 == 0 and (nk_ + 127) // 128 * nzl >= 256 else 1      # hmg_power_batch's tile rule (256 CUs)
    if hint_n is not None and not args.per_pair:
        nconst = hint_n.numpy().view(np.int32)[:B]
        pw_model, pw_tens = power_bytes_moved(nzl, nm_, nk_, npair, nconst, vec)
    else:
        pw_model, pw_tens = 2 * tens_bytes + 8.0 * nzl * nk_ * (2 * npair + 1) + 8.0 * B * 31, 2 * tens_bytes
    alg = {"power": 2 * tens_bytes + 8.0 * nzl * nk_ * (2 * npair + 1) + 8.0 * B * 8,     # SURVEY 8d W_mass, batched
           "nfw": tens_bytes,
           "fft": 2 * 8.0 * B * nxs + 2 * 16.0 * B * (nxs // 2 + 1) + tens_bytes}     # SURVEY 8d W_fft (unfused chain)
    model = {"power": pw_model, "nfw": tens_bytes + 8.0 * B * 40, "fft": tens_bytes + 8.0 * B * 10}
    if tensor:
        model["fft"] += model["nfw"]
        alg["fft"] += alg["nfw"]
    # The ALGORITHMIC bytes of THIS design: every (z,m,k) tensor crosses HBM exactly twice - written
    # once by its producer, read once by the batched mass integrals - plus the per-(z,m) side arrays.  SURVEY 8d's model
    # (`alg`) prices the reference's unfused pipeline: integrand and spectrum arrays of the FFT chain in HBM (here: LDS) and
    # one pass over two tensors per spectrum (here: one pass for all six).
    nq_ = int(h.p["sigma2_numks"])
    nseg_ = (nq_ + 79) // 80
    side = {"front": 8.0 * nzl * nq_ + 2 * 8.0 * nq_ + 8.0 * nseg_ * B + 8.0 * B * (5 + 36 + 7 + 4),     # P(k',z), k', w; partial sums out; c, r_vir, r_s, M/R_200c, series rows, Battaglia rows, occupations
            "nfw": 8.0 * nseg_ * B + 8.0 * B * (3 + 36 + 3),                                            # partial sums in, n, b, sigma2 out; row constants + series rows in
            "fft": 8.0 * B * (7 + 2) + 8.0 * B * 8,                                                     # Battaglia rows in, hint out; coefficient rows out (chain)
            "power": 8.0 * B * (8 + 2) + 8.0 * nzl * nk_ * (2 * npair + 1)}                              # coefficient rows + hints in, P_lin in, 12 spectra out
    design = {"front": side["front"], "nfw": tens_bytes + side["nfw"], "fft": tens_bytes + side["fft"],
              "power": 2 * tens_bytes + side["power"]}
    if tensor:
        design["fft"] += design.pop("nfw")
    design_step = float(sum(design.values()))

    def kernel_entry(key, sub, bound, note):
        ms_ = kms[key]
        moved = pmc_bytes(sub) or model[key]
        e = {"bound": bound, "ms": ms_, "bytes_moved": moved,
             "bytes_source": f"pmc (profiles/{PROFILE_ROUND}/pmc_traffic.json)" if pmc_bytes(sub) else model_source,
             "bytes_model": model[key], "hbm_GBps": moved / (ms_ * 1e-3) / 1e9 if ms_ else None,
             "hbm_frac": moved / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_ else None,
             "survey_alg_bytes": alg[key], "note": note}
        v = valu(sub)
        if v:
            e.update(v)
            e["valu_issue_frac"] = v["valu_issue_bound_ms"] / ms_ if ms_ else None
        return e

    kernels = {
        "power_batch_kernel": kernel_entry("power", KNAME["power"], "hbm",
                                           "fused 1h+2h mass integrals of all 6 spectra, 1 launch/step; the constant "
                                           "left-fill prefix of the Battaglia tensor is substituted, not read"),
        "nfw_kernel": kernel_entry("nfw", KNAME["nfw"], "fp64-valu",
                                   "series / Si-Ci evaluation per point, one 8-B store per point"
                                   + ("; launched as hmg::rows_group_kernel: n(z,m), b(z,m) tiles | Battaglia row "
                                      "parameters | NFW rows in one grid" if grouped else "")),
        "profile_fused_kernel": kernel_entry("fft", KNAME["fft"], "fp64-valu + LDS",
                                             "integrand + in-LDS packed-real FFT + k-interpolation per (z,m) row; "
                                             "HBM traffic = the output row"
                                             + ("; launched as hmg::tensor_group_kernel: per-z chain (sigma^2 -> n, b -> HOD "
                                                "sums -> coefficient rows) | profile rows | analytic NFW rows in one grid: "
                                                "ms, bytes and instruction counts are those of BOTH tensors' rows" if tensor else
                                                "; launched as hmg::profile_group_kernel: per-z chain (HOD sums -> "
                                                "coefficient rows) | profile rows in one grid" if grouped else "")),
    }
    if tensor:
        kernels["nfw_kernel"] = {"bound": "fp64-valu", "ms": None, "launched_as": "hmg::tensor_group_kernel",
                                 "note": "no launch of its own: the analytic NFW rows ride in hmg::tensor_group_kernel (see "
                                         "profile_fused_kernel); HMG_NO_TENSOR_GROUP=1 restores the rows group launch"}
    for key, name in (("power_batch_kernel", "power"), ("nfw_kernel", "nfw"), ("profile_fused_kernel", "fft")):
        if not (tensor and name == "nfw"):
            kernels[key]["launched_as"] = "hmg::" + KNAME[name]
    pw = kernels["power_batch_kernel"]
    # the kernel the step spends most of its time in is VALU/LDS-bound, not HBM-bound: its own roofline block
    dom_key = max(kernels, key=lambda k: kernels[k]["ms"] or 0.0)
    dom = kernels[dom_key]
    roofline_time_dominant = {
        "kernel": dom["launched_as"], "share_of_step": (dom["ms"] or 0.0) / (dt_max / K * 1e3),
        "bound": dom["bound"], "ms_per_launch": dom["ms"],
        "valu_insts_per_wave": dom.get("valu_insts_per_wave"), "valu_issue_bound_ms": dom.get("valu_issue_bound_ms"),
        "valu_issue_frac": dom.get("valu_issue_frac"),
        "hbm_GBps": dom["hbm_GBps"], "hbm_frac": dom["hbm_frac"],
        "note": "valu_issue_frac = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz / measured time (a wave64 fp64 VALU "
                "instruction holds its SIMD-32 for 4 cycles); null when no counter profile of this build is stored"}
    step_bytes = None
    if pmc:
        step_bytes = float(sum(r["hbm_bytes_per_launch_corrected"] * r.get("launches_per_step", 1)
                               for r in pmc["kernels"].values()))
    if step_bytes is None:
        step_bytes = float(sum(model.values()))

    # ---- result hand-over over PCIe (never part of `value`): one device block -> one pinned host block
    pcie = None
    if world == 1:
        blk = h.spectra_block(PAIRS)
        blk.compute(); blk.fetch()                                  # allocations, first touch
        t1 = time.perf_counter()
        for _ in range(5):
            blk.fetch()
        d2h_ms = (time.perf_counter() - t1) / 5 * 1e3
        ins = [h.Pzk, h.sPzk, zs, ms, ks]
        t1 = time.perf_counter()
        for _ in range(5):
            keep = [ctx.upload(a) for a in ins]
        ctx.sync()
        h2d_ms = (time.perf_counter() - t1) / 5 * 1e3
        # the same inputs staged in page-locked memory: asynchronous DMAs at link speed, no bounce copies
        pins = [nat.PinnedArray(ctx, a.shape) for a in ins]
        for pa, a in zip(pins, ins):
            pa.array[...] = a
        dsts = [ctx.empty(a.shape) for a in ins]
        for d_, pa in zip(dsts, pins):
            ctx.copy_from_pinned(d_, pa)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            for d_, pa in zip(dsts, pins):
                ctx.copy_from_pinned(d_, pa)
        ctx.sync()
        h2d_pinned_ms = (time.perf_counter() - t1) / 5 * 1e3
        assert all(np.array_equal(d_.numpy(), a) for d_, a in zip(dsts, ins))

        # ---- streamed hand-over: a double-buffered result block; pass i's copy to the host runs on the copy lane
        # under pass i+1's kernels, the host consumes pass i-1 meanwhile.  Same kernels, same numbers.
        streamed = None
        if can_both:
            blk2 = h.spectra_block(PAIRS, nbuf=2)

            def pass_into(slot):
                h.init_mass_function(ms)
                h.add_nfw_profile("nfw", ignore_existing=True)
                h.add_battaglia_profile("electron", family="AGN", xmax=args.xmax, nxs=args.nxs, ignore_existing=True)
                h.add_hod("g", mthresh=mthr, ignore_existing=True)
                blk2.compute(slot)
            for slot in (0, 1):
                pass_into(slot)
            ctx.sync()
            g2 = [ctx.capture(lambda slot=slot: pass_into(slot)) for slot in (0, 1)]
            consumed = [0.0]

            def stream(nsteps):
                for s_ in range(nsteps):
                    slot = s_ % 2
                    if s_ >= 2:
                        ctx.wait(blk2._EV_DONE + slot)          # device-side: the block's last copy has left
                    ctx.replay(g2[slot])
                    blk2.fetch_async(slot)
                    if s_ >= 1:                                  # the host reads pass s-1 while pass s runs
                        got = blk2.wait((s_ - 1) % 2)
                        consumed[0] += float(got[PAIRS[0]][0][0, 0])
                got = blk2.wait((nsteps - 1) % 2)
                return got
            stream(8)
            ctx.sync()
            t1 = time.perf_counter()
            last = stream(K)
            ctx.sync()
            st_ms = (time.perf_counter() - t1) / K * 1e3
            res_now = spec.results()
            same = all(np.array_equal(last[p_][i], res_now[p_][i]) for p_ in PAIRS for i in (0, 1))
            streamed = {"ms_per_step_incl_results": st_ms, "bit_equal_to_timed_outputs": bool(same),
                        "note": "K passes, each computing into one of two device result blocks and copied to its pinned "
                                "twin on the copy lane behind an event; the host waits for and reads pass i-1 while pass "
                                "i runs (hmg_event_synchronize on that copy only)"}
        pcie = {"d2h_results_ms": d2h_ms, "h2d_inputs_ms": h2d_ms, "h2d_inputs_pinned_ms": h2d_pinned_ms,
                "results_MB": blk.nbytes / 1e6,
                "inputs_MB": sum(a.nbytes for a in ins) / 1e6,
                "d2h_GBps": blk.nbytes / d2h_ms / 1e6,
                "ms_per_step_incl_transfers": dt_max / K * 1e3 + d2h_ms + h2d_ms,
                "streamed": streamed,
                "note": "results: 12 (nz,nk) spectra in one device block, one asynchronous copy into one pinned host "
                        "block, numpy views handed out (HaloModel.spectra_block); inputs: pageable numpy uploads "
                        "(h2d_inputs_ms) and the same arrays staged in pinned memory (h2d_inputs_pinned_ms). "
                        "ms_per_step_incl_transfers serialises step + copies; `streamed` overlaps the result copy with "
                        "the next pass"}
        del keep

    # ---- Config 5: Limber C_kk + C_kg at 2000 multipoles on the gathered, device-resident spectra
    limber = None
    if not args.no_limber:
        full = hm.Cosmology(dict(h.p), accuracy="low", engine="analytic")
        full.ctx = ctx
        ells = np.linspace(100, 6000, 2000)
        iP = {p: i for i, p in enumerate(PAIRS)}
        def dev_pair(pair):
            i = iP[pair]
            return (spec.full[2 * i], spec.full[2 * i + 1])       # (P_1h, P_2h), summed inside the kernel
        run = lambda: (full.C_kk(ells, zs, ks, dev_pair(("nfw", "nfw")), lzs1=2.5, lzs2=2.5),       # noqa: E731
                       full.C_kg(ells, zs, ks, dev_pair(("g", "nfw")), gzs=0.8, lzs=2.5))
        run()                                                      # uploads of windows, first launch
        ctx.sync()
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
            ckk, ckg = run()
        lim_ms = (time.perf_counter() - t1) / reps * 1e3
        limber = {"ells": 2000, "C_kk+C_kg_ms": lim_ms,
                  "config5_ms_end_to_end": dt_max / K * 1e3 + lim_ms,
                  "C_kk[0]": float(ckk[0]), "C_kg[0]": float(ckg[0]),
                  "note": "host wall per (C_kk, C_kg) pair incl. the window functions on the host, their upload and "
                          "the D2H of C_ell; P(z,k) stays in HBM.  config5 = one step + this"}

    pts = npair * zs.size * ms.size * ks.size
    out = {
        "metric": "(z,m,k) grid-points/sec for P_1h+P_2h",
        "value": pts * K / dt_max, "unit": "grid-points/s",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "transport": "rccl" if rccl_ranks else None, "steps": K, "warmup": W, "ms_per_step": dt_max / K * 1e3,
        "host_issue_ms_per_step": t_issue / K * 1e3, "preconditioning_steps": PRECONDITION,
        "launch_mode": ("hip-graph replay" if use_graph else "eager launches from a recorded call list" if args.call_list
                        else "eager launches") + (", lanes" if args.lanes else ""),
        "launch_modes": {"timed": {"mode": "hip-graph replay" if use_graph else "eager launches from a recorded call list"
                                           if args.call_list else "eager launches", "ms_per_step": dt_max / K * 1e3},
                         "other": other_mode,
                         "note": "same launches either way; `value` is the timed mode.  Graph replays leave a gap between "
                                 "consecutive launches of the graph (profiles/r06/slab4_timeline.txt)"},
        "kernel_events": f"HIP events around the three large kernels on every {BRACKET_EVERY}th timed step"
                         + (" (graph event nodes)" if use_graph and events_in_graph else
                            " (eager launches from a recorded call list)" if t_brack is not None else " (eager launches)"),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{'Config 3' if (args.nxs, args.xmax) == (5000, 20.0) else 'Config-3 grid, radial grid of the reference callers (examples/lensing_baryons.py:27)'}: zs={zs.size} ms={ms.size} ks={ks.size}, analytic NFW + "
                               f"Battaglia AGN electron (nxs={args.nxs}, xmax={args.xmax:g}) + HOD(mthresh=10^10.5), "
                               f"6 auto/cross spectra 1h+2h, full path per step",
                   "parallelism": f"z-slab x{world}" if world > 1 else "single GPU",
                   "grid_points_per_step": pts},
        "roofline": {"kernel": "hmg::power_batch_kernel" if not args.per_pair else "hmg::power_kernel x6 (per-pair path)",
                     "bound": "hbm", "achieved": pw["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": pw["hbm_frac"], "traffic": pmc_bytes("power_batch_kernel"),
                     "traffic_source": (f"stored profile profiles/{PROFILE_ROUND}/pmc_traffic.json of this build: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                        "separate passes, (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch "
                                        "(gfx950 FETCH_SIZE correction)") if pmc_bytes("power_batch_kernel") else model_source,
                     "bytes_moved": pw["bytes_moved"], "bytes_source": pw["bytes_source"], "bytes_model": pw["bytes_model"],
                     "ms_per_launch": kms["power"],
                     "design_alg_bytes_per_launch": design["power"],
                     "note": f"HBM-bound kernel ({100 * (kms['power'] or 0.0) / (dt_max / K * 1e3):.0f} % of the step); time-dominant: "
                             f"{roofline_time_dominant['kernel']} ({100 * roofline_time_dominant['share_of_step']:.0f} % of the step, "
                             f"{roofline_time_dominant['bound']}, VALU issue frac {roofline_time_dominant['valu_issue_frac']}) "
                             "- its block is roofline_time_dominant.  achieved/frac use the bytes the launch actually moves (counter bytes when a profile of this "
                             "build and configuration is stored, else the launch's own skip rule evaluated on the hint array: see "
                             "traffic_source / bytes_source); design_alg_bytes_per_launch = read both tensors once + side arrays "
                             "(the hinted constant prefix of the Battaglia tensor is substituted, not read, which is why "
                             "bytes_moved is below it).  SURVEY 8d's byte model of the UNFUSED reference pipeline is in "
                             "survey_model, outside this block: it is not a bandwidth of this design"},
        "roofline_time_dominant": roofline_time_dominant,
        "kernels": kernels,
        "launches_per_step": (ctx.graph_kernel_nodes(g_count) if g_count is not None else None),
        "launches_per_step_source": ("kernel nodes of the captured step (hipGraphGetNodes), counted by the library at capture end"
                                     if g_count is not None else "not counted: eager launches (--no-graph / --stages)"),
        "launch_grouping": ("front (sigma^2 contraction | halo stage | HOD occupations), rows group, profile group, "
                            "mass integrals" if grouped else "one launch per stage (HMG_NO_GROUPS=1)"),
        "kernel_source_sha16": sha, "stale_profiles_ignored": stale or None,
        "brackets_missing": brackets_missing or None,
        "brackets_missing_note": ("nfw: by design - the NFW rows have no launch of their own (tensor group)" if tensor else None),
        "step_hbm_bytes": step_bytes,
        "step_hbm_bytes_source": (f"pmc (profiles/{PROFILE_ROUND}/pmc_traffic.json, all kernels of a step)" if pmc else model_source),
        "step_hbm_frac": step_bytes / (dt_max / K) / 1e9 / HBM_PEAK_GBS,
        "design_alg_bytes_per_step": design_step,
        "design_alg_bytes": {"tensor_passes": 4, "tensor_bytes": tens_bytes, "side_arrays": float(sum(side.values())),
                             "per_launch": design,
                             "note": "2 tensors written once (NFW rows, profile rows) + read once (one pass of the mass integrals "
                                     "for all six spectra) + per-(z,m) side arrays; DESIGN.md section 4"},
        "step_hbm_frac_design": design_step / (dt_max / K) / 1e9 / HBM_PEAK_GBS,
        "survey_model": {
            "survey_alg_bytes_per_step": float(sum(alg.values()) + (sum([1, 1, 1, 2, 1, 2]) - 2) * tens_bytes),
            "survey_alg_bytes_mass_integrals": alg["power"],
            "survey_bytes_per_s_equiv": (float(sum(alg.values()) + (sum([1, 1, 1, 2, 1, 2]) - 2) * tens_bytes) / (dt_max / K) / 1e9),
            "unit": "GB/s-equivalent",
            "note": "SURVEY 8d prices the reference's UNFUSED pipeline (FFT chain through HBM 3.16 GB, one pass over two tensors "
                    "per spectrum 4.30 GB).  This design removes that traffic (FFT chain in LDS; all six pairs in one pass; "
                    "the constant left-fill prefix substituted from a hint), so this figure exceeds the HBM peak: it is a "
                    "speed-up equivalent, NOT a bandwidth, and no roofline fraction is formed from it"},
    }
    if gather_ms is not None:
        out["gather_ms_per_step"] = gather_ms["max_over_ranks_ms"]
        out["gather"] = gather_ms
    if args.stages and stage_ms:
        out["stages_ms"] = dict(zip(["mass_function", "nfw", "battaglia_fft", "hod", "spectra"],
                                    np.mean(np.array(stage_ms), axis=0).tolist()))
    if world == 1 and not args.no_readme:
        out["readme_config2"] = readme_config2(ctx, with_cpu=not args.no_cpu_baseline)
    if world == 1 and not args.no_long_grid and (args.nxs, args.xmax) == (5000, 20.0):
        out["long_grid"] = long_grid_block(ctx, zs, ms, ks, mthr, PAIRS)
    if pcie is not None:
        out["pcie"] = pcie
    if limber is not None:
        out["limber"] = limber
    if cpu is not None:
        cb, sel, ref = cpu
        res = spec.results()
        worst = 0.0
        for (a, b), R in zip(PAIRS, ref):
            P = (res[(a, b)][0] + res[(a, b)][1])[sel]
            tol = 1e-8 * np.abs(R) + 1e-12 * np.max(np.abs(R), axis=-1, keepdims=True)
            worst = max(worst, float(np.max(np.abs(P - R) / tol)))
        cb["parity_worst_dP_over_tol"] = worst
        out["cpu_baseline"] = cb
    print(json.dumps(out))
    comm.barrier()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
