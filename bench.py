#!/usr/bin/env python3
"""Headline benchmark: postage-stamps/s of the IMCOM stamp path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (A build, B build, blocked Cholesky, T solve, U/C-Sigma-kappa maps,
coaddition epilogue) over one batch of synthetic postage stamps that are already resident in HBM.  (Since round 2 the
single-kappa maps ride in the solve launches: the forward launches leave the column sums of Y^2, the backward ones T in
float32 and the sums of X^2, so the "finalize" stage is a reduction of a few MB and the solve family carries that work.)
Workload = BASELINE.json configs[1] ("cfg2": 48x48-output stamps, 6 exposures, analytic Roman-like
PSF, fp64, Cholesky kappa/C = 6e-4).  One process per GPU, no data-path collective
(blocks/stamps are independent): weak scaling, each rank runs its own batch.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime initialises (pyimcom_amd/__init__.py has the reason)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak (spec); 77.2 measured with tools/mfma_f64_bench
CONFIG_INDEX = {"cfg1": 0, "cfg2": 1, "cfg2f": 1, "cfg3": 2, "cfg4": 3, "cfg5": 4}  # BASELINE.json configs[] of each workload


class Telemetry:
    """Shader / memory clock, board power and temperatures of the GPU this rank runs on, sampled from sysfs (amdgpu hwmon: freq1 = sclk,
    freq2 = mclk, power1_input, temp2 = junction, temp3 = memory) by a thread while a timed region runs -- what tells a reader of the
    line why this box is faster or slower than another one (boxes of the pool differ by ~6 % on the same kernel).  The card is found by
    the PCI address torch reports for the device; anything that cannot be read is left out (never an error)."""

    def __init__(self, device_index=0, period_s=0.05):
        import glob

        self.dir, self.period, self.rows, self._stop, self._thr = None, period_s, [], None, None
        try:
            import torch

            pr = torch.cuda.get_device_properties(device_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for c in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(c).endswith(addr):
                    h = glob.glob(os.path.join(c, "hwmon", "hwmon*"))
                    if h:
                        self.dir, self.addr = h[0], addr
        except Exception:  # noqa: BLE001
            pass

    def _read(self, name, scale):
        try:
            return float(open(os.path.join(self.dir, name)).read()) * scale
        except Exception:  # noqa: BLE001
            return None

    def sample(self):
        return (self._read("freq1_input", 1e-6), self._read("freq2_input", 1e-6), self._read("power1_input", 1e-6),
                self._read("temp2_input", 1e-3), self._read("temp3_input", 1e-3))

    def start(self):
        import threading

        if self.dir is None:
            return self
        self.rows, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self.rows.append(self.sample())
                self._stop.wait(self.period)

        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is None:
            return {"source": None}
        self._stop.set()
        self._thr.join()
        out = {"source": f"sysfs hwmon of {self.addr}, every {self.period * 1e3:.0f} ms", "samples": len(self.rows)}
        for k, name in enumerate(("sclk_mhz", "mclk_mhz", "power_w", "temp_junction_c", "temp_mem_c")):
            v = sorted(r[k] for r in self.rows if r[k] is not None)
            if v:
                out[name] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        cap = self._read("power1_cap", 1e-6)
        if cap is not None:
            out["power_cap_w"] = cap
        return out


def spread(ms):
    """min / median / max of a list of per-step milliseconds."""
    v = sorted(ms)
    return {"min": v[0], "median": v[len(v) // 2], "max": v[-1]} if v else None


# LDS-DMA gathers of table rows served by the XCDs' L2s: 16.8-18.8 TB/s chip-wide (MI355X_MICROARCH.md, "Indexed rows: gather into LDS")
L2_GATHER_PEAK_GBS = 18800.0


def on_table_samples(batch, cfg):
    """Samples (i <= j) of every stamp's symmetric half whose stencil lies ON the pair's overlap table -- the ones build_A_kernel stages and
    interpolates; the others are left zero by the reference (psfutil.py:1691-1702: xi < 4 or xi >= ng - 5 ...) and by the kernel, and
    are no work.  Counted on the device from the batch's own pixel lists with the kernel's test (csrc/build_a.hip: dx / dscale + nc + 6
    truncated to a cell); at cfg-2 every sample is on its table, at the reference's benchmark shape 11 % are not."""
    import numpy as np
    import torch

    ng, nc, ds = cfg.nsamp + 12, float(cfg.nc), float(cfg.dscale)
    out = np.zeros(batch.batch, dtype=np.float64)
    for s in range(batch.batch):
        n = int(batch.n[s])
        x, y = batch.x[s, :n], batch.y[s, :n]
        tot = 0
        step = max(1, (1 << 24) // max(n, 1))
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            cx = torch.floor((x[r0:r1, None] - x[None, :]) / ds + nc + 6.0)
            cy = torch.floor((y[r0:r1, None] - y[None, :]) / ds + nc + 6.0)
            on = (cx >= 4) & (cx < ng - 5) & (cy >= 4) & (cy < ng - 5)
            on &= torch.arange(n, device=x.device)[None, :] >= torch.arange(r0, r1, device=x.device)[:, None]
            tot += int(on.sum())
        out[s] = tot
    return out


def roofline_build_A(n_arr, ms_per_step, traffic_bytes=None, traffic_src=None, samples_on_table=None):
    """The A builder against the roofs that bound it ("a table-gather + FMA kernel -> LDS / L2 gather bandwidth and vector
    fp64; report HBM GB/s").  Per sample of the symmetric half that lies ON its overlap table (``samples_on_table`` per stamp, from
    on_table_samples; default: all N (N + 1) / 2) the kernel stages the ten stencil rows as 5 x 16-byte LDS-DMA pieces each = 800 B from
    L1 / L2 into LDS and computes 330 flops; every entry of A is written once (8 N^2 bytes per stamp).  SURVEY 8(d)'s own roofs are
    ``hbm_frac`` (bytes written over 8 TB/s) and ``valu_frac`` (flops over the vector fp64 peak); ``frac`` is the staged bytes per second
    over the guide's MEASURED rate for LDS-DMA row gathers out of L2 -- an empirical gather rate, not a bound, named so in the line."""
    import numpy as np

    n = np.asarray(n_arr, dtype=np.float64)
    half = float((n * (n + 1) / 2).sum())
    samples = half if samples_on_table is None else float(np.asarray(samples_on_table, dtype=np.float64).sum())
    t = ms_per_step * 1e-3
    staged = samples * 800.0 / t / 1e9
    return {"kernel": "build_A_kernel", "bound": "l2-gather", "achieved": staged, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": staged / L2_GATHER_PEAK_GBS,
            "peak_is": "empirical gather rate (guide: LDS-DMA row gathers out of L2, measured), not a bound",
            "samples_per_s": samples / t, "samples_on_table_share": samples / half if half else None, "staged_bytes_per_sample": 800, "avg_launch_ms": ms_per_step,
            "hbm_write_GBs": float((8.0 * n * n).sum()) / t / 1e9, "hbm_frac": float((8.0 * n * n).sum()) / t / 8e12,
            "valu_TFLOPs": 330.0 * samples / t / 1e12, "valu_frac": 330.0 * samples / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "traffic": traffic_bytes, "traffic_source": traffic_src,
            "note": "achieved = 800 B staged per on-table sample of the symmetric half over the launch time; SURVEY 8(d)'s roofs are hbm_frac and valu_frac"}


def roofline_chol(n_arr, factorisations, ms_gemm, ms_diag, launches):
    """The blocked Cholesky (chol_update + chol_trsm + chol_diag launches) against the fp64 matrix peak: N^3 / 3 per factorisation."""
    import numpy as np

    n = np.asarray(n_arr, dtype=np.float64)
    flops = float((n**3 / 3.0).sum()) * factorisations
    t = (ms_gemm + ms_diag) * 1e-3
    ach = flops / t / 1e12 if t > 0 else 0.0
    return {"kernel": "chol_update_kernel+chol_trsm_kernel+chol_diag_kernel", "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / FP64_MFMA_PEAK_TFLOPS, "factorisations_per_step": factorisations, "ms_per_step": ms_gemm + ms_diag, "ms_diag_blocks": ms_diag,
            "launches": launches, "traffic": None, "note": "algorithmic N^3/3 per factorisation; the update launches execute the diagonal tiles in full (+13 % at N = 2.2k)"}


def pmc_kernel_traffic(kernel, batch, cfg_name, pattern="r*_pmc_traffic.json"):
    """HBM bytes per launch of `kernel` from the newest committed counter passes of the headline command (as pmc_traffic); ``pattern``:
    another leg's passes (r*_pmc_traffic_iter.json: tools/pmc_iter_traffic.sh)."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pattern)))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    from pyimcom_amd._lib import source_sha16

    if doc.get("batch") != batch or doc.get("workload") != cfg_name or doc.get("csrc_sha16") != source_sha16():
        return None, None
    ks = [v for k, v in doc["kernels"].items() if k.split("<")[0].split("(")[0] == kernel]
    n = sum(k["launches"] for k in ks)
    return (sum(k["traffic_bytes_per_launch"] * k["launches"] for k in ks) / n if n else None), os.path.join("profiles", os.path.basename(files[-1]))


def cpu_baseline(cfg, stamps, psfs, target, budget_s=20.0, modes=("all", "one", "processes")):
    """The oracle (CPU restatement of the reference path, oracle/) timed on this box's host cores on a bounded sample of
    the same workload: whole stamps -- A and B build (C interpolators, OpenMP over the samples), LAPACK potrf / potrs (numpy
    / scipy BLAS threads), maps, coaddition -- first on all cores, then on ONE thread (threadpoolctl + orc.set_threads),
    each for about half the budget, with the split build / solve / epilogue that SURVEY 8(d) asks for."""
    import numpy as np

    from oracle import oracle as orc

    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # pragma: no cover
        threadpool_limits = None

    g, tabs, C = orc.stamp_tables(cfg, psfs, target)
    C = float(C[0])  # one target PSF
    E = psfs.shape[0]
    tri = lambda i, j: (2 * E - i + 1) * i // 2 + j - i
    tab = np.zeros((E, E), np.int32)
    pen = np.zeros((E, E))
    for a in range(E):
        for b in range(E):
            tab[a, b] = tri(a, b) if a <= b else (tri(b, a) | (1 << 30))
            pen[a, b] = -cfg.flat_penalty / E + (cfg.flat_penalty if a == b else 0.0)
    io = np.arange(E) + E * (E + 1) // 2

    def sample(budget, nthreads):
        orc.set_threads(nthreads)
        stages, done = {}, 0
        t0 = time.perf_counter()
        for st in stamps:
            orc.stamp_full(cfg, g, tabs, C, st, tab, pen, io, timings=stages)
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        return done, dt, {k: v / done * 1e3 for k, v in stages.items()}

    cores = os.cpu_count()
    blas = "default"
    try:  # threads the BLAS / LAPACK behind numpy and scipy will use (threadpoolctl reads the loaded libraries)
        from threadpoolctl import threadpool_info

        blas = ", ".join(f"{p_.get('internal_api', p_.get('user_api'))} {p_.get('num_threads')} threads" for p_ in threadpool_info() if p_.get("user_api") == "blas") or "default"
    except Exception:  # pragma: no cover
        pass
    # the C interpolators are called once per exposure pair (~80k samples): more than ~32 OpenMP threads only add
    # start-up and spinning next to the BLAS threads (measured: 256 threads 5.7 s per stamp, 1 thread 0.3 s)
    omp = min(cores, 32)
    done, dt, stages = sample(budget_s * (0.5 if "one" in modes else 1.0), omp)
    if threadpool_limits is not None and "one" in modes:
        with threadpool_limits(limits=1):
            done1, dt1, stages1 = sample(budget_s * 0.5, 1)
    else:
        done1, dt1, stages1 = 0, 0.0, {}
    orc.set_threads(0)
    out = {
        "value": done / dt,
        "unit": "postage-stamps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{done} whole {cfg.name} stamps (N~{stamps[0].n}, m={cfg.m}) through oracle/ on {cores} cores ({dt:.1f} s): C interpolators "
                  f"with {omp} OpenMP threads, " + ("numpy conjugate gradients per output pixel" if cfg.kernel == "Iterative" else "scipy potrf/potrs") + f" on {blas}"
                  + (f"; then {done1} stamps on 1 thread ({dt1:.1f} s)" if done1 else ""),
        "blas": blas,
        "stage_ms_per_stamp": stages,
    }
    if done1:
        out["one_thread"] = {"value": done1 / dt1, "unit": "postage-stamps/s", "cores": 1, "stage_ms_per_stamp": stages1}
    if "processes" in modes:
        out["processes"] = cpu_processes(cfg.name, cores, budget_s * 0.5)
    return out


def cpu_worker(cfg_name, budget_s):
    """One single-threaded oracle process of cpu_processes(): whole stamps of the workload for `budget_s` seconds -> one JSON line."""
    import numpy as np

    from oracle import oracle as orc
    from pyimcom_amd import synth

    try:
        from threadpoolctl import threadpool_limits

        limit = threadpool_limits(limits=1)
    except ImportError:  # pragma: no cover
        limit = None
    orc.set_threads(1)
    cfg = synth.CONFIGS[cfg_name]
    first = int(os.environ.get("IMCOM_CPU_WORKER_FIRST", "0"))
    E = cfg.n_expo if isinstance(cfg.n_expo, int) else cfg.n_expo[1]
    psfs, target = synth.make_psfs(cfg, E)
    g, tabs, C = orc.stamp_tables(cfg, psfs, target)
    tri = lambda i, j: (2 * E - i + 1) * i // 2 + j - i
    tab = np.array([[tri(a, b) if a <= b else (tri(b, a) | (1 << 30)) for b in range(E)] for a in range(E)], dtype=np.int32)
    pen = np.array([[-cfg.flat_penalty / E + (cfg.flat_penalty if a == b else 0.0) for b in range(E)] for a in range(E)])
    io = np.arange(E) + E * (E + 1) // 2
    print("ready", flush=True)
    sys.stdin.readline()  # every worker starts its clock at the parent's signal
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        orc.stamp_full(cfg, g, tabs, float(C[0]), synth.make_stamp(cfg, first + done), tab, pen, io)
        done += 1
    print(json.dumps({"done": done, "seconds": time.perf_counter() - t0}), flush=True)
    del limit
    return 0


def cpu_processes(cfg_name, cores, budget_s):
    """The reference's own way of using a many-core host is one process per block (docs/run_README.rst:81-100): P single-threaded
    oracle processes side by side, each coadding whole stamps for `budget_s` seconds (started as child processes; tables are built
    before the clock starts).  P = half the logical cores, at most 64 (memory: ~0.4 GB per process at cfg-2)."""
    import subprocess

    P = max(1, min(64, cores // 2))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = []
    for k in range(P):
        e = dict(env, IMCOM_CPU_WORKER_FIRST=str(1000 * k))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--config", cfg_name, "--cpu-worker", str(budget_s)], env=e, cwd=ROOT,
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
    try:
        for p_ in procs:
            assert p_.stdout.readline().strip() == "ready"
        t0 = time.perf_counter()
        for p_ in procs:
            p_.stdin.write("go\n")
            p_.stdin.flush()
        res = [json.loads(p_.stdout.readline()) for p_ in procs]
        wall = time.perf_counter() - t0
    except Exception as exc:  # a worker that died: report, do not fail the bench line
        for p_ in procs:
            p_.kill()
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        for p_ in procs:
            try:
                p_.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p_.kill()
    done = sum(r["done"] for r in res)
    return {"value": done / wall, "unit": "postage-stamps/s", "processes": P, "threads_per_process": 1, "cores": P,
            "sample": f"{done} whole {cfg_name} stamps by {P} single-threaded oracle processes in {wall:.1f} s (one process per block is how the reference "
                      "uses a many-core host)"}


def pmc_traffic(batch, cfg_name):
    """HBM bytes per launch of the solve kernels from the newest committed rocprofv3 --pmc passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this very command, corrected as MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_traffic.py).
    Counters cannot be read from inside the process: the figure is reported with the file it came from, and only when
    that measurement was taken on the same workload and batch AND on the same kernel sources (the file carries the SHA-256
    of csrc/ at measurement time); otherwise null."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    from pyimcom_amd._lib import source_sha16

    if doc.get("batch") != batch or doc.get("workload") != cfg_name or doc.get("csrc_sha16") != source_sha16():
        return None, None  # another workload, or counters taken on other kernel sources than the ones this run was built from
    ks = [v for k, v in doc["kernels"].items() if k.split("<")[0] in ("solve_fwd_kernel", "solve_bwd_kernel")]  # (solve_bwd_kernel<false>: a template since round 4)
    n = sum(k["launches"] for k in ks)
    return (sum(k["traffic_bytes_per_launch"] * k["launches"] for k in ks) / n if n else None), os.path.join("profiles", os.path.basename(files[-1]))


def pmc_band_traffic(kernel="symv4_kernel"):
    """HBM bytes per launch AND matrix of one of the band reduction's kernels from the newest committed counter passes
    (tools/pmc_band_reduce.sh: one reduction of 8 matrices of 2938 rows, FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
    correction as for pmc_traffic) -- only when they were taken on the kernel sources this run was built from; else None."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_band_reduce.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    from pyimcom_amd._lib import source_sha16

    if doc.get("csrc_sha16") != source_sha16() or kernel not in doc.get("kernels", {}):
        return None, None
    return doc["kernels"][kernel]["traffic_bytes_per_launch"] / float(doc["batch"]), os.path.join("profiles", os.path.basename(files[-1]))


def block_workload(dev, n1P, identical=False, seed=5, config="cfg2"):
    """Synthetic block at cfg-2 geometry for the block leg (and for tests/test_gpu_bigblock.py's check of it): the InStamp
    pool of (n1P + 2)^2 InStamps, and per 2 x 2 group of InStamps PSF images [E, ns + 16, ns + 16] (a smooth modulation of the
    analytic PSFs, zero padded; ``identical``: the same images for every group) with their sampling positions yxco [E, 2, ns, ns]
    (a small rotation per exposure, psfutil.py:751-771; none with ``identical``), all resident on the device."""
    import numpy as np
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.select import InStampPool

    cfg = synth.CONFIGS[config]
    E = cfg.n_expo
    inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(seed))
    pool = InStampPool(inst, cfg.n_inframe, device=dev)
    psfs, target = synth.make_psfs(cfg, E)
    ng = (n1P + 3) // 2
    ns = psfs.shape[-1]
    lin = torch.arange(ns, dtype=torch.float64, device=dev) - ns // 2
    base = torch.as_tensor(psfs, device=dev)
    lt = torch.arange(ns, dtype=torch.float64, device=dev) - (ns - 1) / 2.0
    yo, xo = torch.meshgrid(lt, lt, indexing="ij")
    img_all = torch.zeros((ng * ng, E, ns + 16, ns + 16), dtype=torch.float64, device=dev)
    yxco_all = torch.empty((ng * ng, E, 2, ns, ns), dtype=torch.float64, device=dev)
    groups, counts = {}, {}
    for gj in range(ng):
        for gi in range(ng):
            q = gj * ng + gi
            if identical:
                mod = torch.ones((1, 1, 1), dtype=torch.float64, device=dev)
                th = torch.zeros(E, dtype=torch.float64, device=dev)
            else:
                mod = 1.0 + 0.02 * torch.sin(0.05 * lin * (1 + gi % 3))[None, None, :] + 0.02 * torch.cos(0.04 * lin * (1 + gj % 3))[None, :, None]
                th = torch.as_tensor([0.004 * (e - E / 2) + 0.002 * ((gi - gj) % 7 - 3) for e in range(E)], dtype=torch.float64, device=dev)
            img_all[q, :, 8 : 8 + ns, 8 : 8 + ns] = base * mod
            c, sn = torch.cos(th)[:, None, None], torch.sin(th)[:, None, None]
            yxco_all[q] = torch.stack([c * yo + sn * xo, -sn * yo + c * xo], dim=1)
            groups[(gj, gi)] = None  # sampled by the bulk provider
            counts[(gj, gi)] = E
    return cfg, inst, pool, psfs, target, groups, counts, img_all, yxco_all


def block_leg(ctx, dev, n1P=48, reps=3, config="cfg2", warm=16):
    """The same path one level up, as a block of the reference runs it (coadd.py:2003-2084): cfg-2 geometry, ONE block of
    n1P x n1P = 48 x 48 output stamps ("one block = 48 x 48 stamps") whose PSFs change from one 2 x 2 group of
    InStamps to the next (SysMatA.ji_st2psf, psfutil.py:1803-1824: 625 groups, ~100 k overlap tables), and everything the
    headline loop leaves out inside the timed region: the batch plan (2-D tiles of cells against the table arena), PSF sampling
    onto the PSFGrp grid (PSFGrp._sample_psf + normalisation, psfutil.py:709-795, 650-656), spectra and overlap tables per group
    (self / cross / input-output sets, least recently used sets replaced), pixel selection from the InStamp pool, per-stamp pair
    maps, A, B, the LA kernel of the configuration, coaddition, block-map accumulation and boundary recovery.  Outside: the upload
    of the InStamp pool, of the PSF images and of their sampling positions, and the allocation of the two arenas (tables,
    spectra), which a block driver keeps from block to block (BlockTables.reset).  ``reps`` timed blocks: the MEDIAN is reported
    (all of them listed).  ``config="cfg3"``: BASELINE configs[2] "batched across one block" -- the Eigen kernel with its kappa sweep."""
    import numpy as np
    import torch

    from pyimcom_amd import psfs as psfmod
    from pyimcom_amd.blockrun import coadd_block, plan_block
    from pyimcom_amd.stamps import BlockTables

    cfg, inst, pool, psfs, target, groups, counts, img_all, yxco_all = block_workload(dev, n1P, config=config)
    E, ns = cfg.n_expo, psfs.shape[-1]
    order = {k: q for q, k in enumerate(groups)}

    def sample_groups(keys):
        # the groups a batch of stamps needs for the first time, sampled in one call
        q0, q1 = order[keys[0]], order[keys[-1]] + 1
        if [order[k] for k in keys] == list(range(q0, q1)):
            im, yx = img_all[q0:q1], yxco_all[q0:q1]
        else:
            idx = torch.tensor([order[k] for k in keys]).pin_memory().to(dev, non_blocking=True)
            im, yx = img_all[idx], yxco_all[idx]
        return psfmod.sample_psf(im.reshape(-1, ns + 16, ns + 16), ns, yx.reshape(-1, 2, ns, ns), psf_norm=True, ctx=ctx)

    fams = ("psf_sample", "psf_spectra", "psf_overlap", "select", "build_A", "build_B", "chol_gemm", "chol_diag", "solve_gemm", "finalize",
            "eigen_trd", "eigen_applyq", "lakernel1", "eigen_gemm", "epilogue", "block_acc")
    from pyimcom_amd.blockrun import memory_plan

    mplan = memory_plan(cfg, pool, n1P, E, ctx=ctx)  # stamps first, the arenas get the rest (up to every table of the block)
    tabs = BlockTables(groups, target, cfg.nfft, ctx=ctx, device=dev, group_count=counts, bulk_provider=sample_groups, cells=True,
                       capacity=mplan["capacity"], spec_capacity=mplan["spec_capacity"],
                       eager_groups=os.environ.get("BENCH_EAGER_GROUPS", "1") != "0")  # (A/B runs: BENCH_EAGER_GROUPS=0)

    def one():
        tabs.reset()  # table construction (PSF sampling, spectra, overlap tables) is part of the block
        return coadd_block(cfg, pool, tabs, n1P, E)

    # warm-up (kernels loaded, workspace and per-batch buffers grown to what a pass of this block takes: a first pass that has to
    # allocate tens of GB does so with the GPU idle), then the timed block(s): a corner of the block, and the first pass of the plan
    # ONE warm-up pass: the first pass of the plan (kernels loaded, the workspace tensor and the per-batch buffers at the size the block's
    # passes take).  The plan is made from torch's allocator alone and counts what the warm-up leaves behind as available: it is the same
    # plan before and after (checked below: a plan that moved gets a second warm-up pass; until round 5 the library's own hipMalloc beside
    # torch's cache made it depend on history).
    first = plan_block(cfg, pool, tabs, n1P)
    coadd_block(cfg, pool, tabs, n1P, E, chunks=first[:1], pad_sides=None)
    torch.cuda.synchronize()
    again = plan_block(cfg, pool, tabs, n1P)
    plan_stable = [len(c) for c in again] == [len(c) for c in first]
    if not plan_stable:  # (an unstable plan would be timed with a pass size that was never warmed up) one more warm-up pass with the new plan
        coadd_block(cfg, pool, tabs, n1P, E, chunks=again[:1], pad_sides=None)
        torch.cuda.synchronize()
        plan_stable = [len(c) for c in plan_block(cfg, pool, tabs, n1P)] == [len(c) for c in again]
    ctx.profile_enable(True)
    runs = []
    for _ in range(reps):
        ctx.profile_reset()
        t0 = time.perf_counter()
        maps = one()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        runs.append((dt, {f: ctx.profile_get(f)[0] for f in fams}))
    ctx.profile_enable(False)
    dt, stages = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    stages = {f: v for f, v in stages.items() if v > 0}
    chunks = plan_block(cfg, pool, tabs, n1P)
    return {
        "value": n1P * n1P / dt, "unit": "postage-stamps/s", "ms_per_block": dt * 1e3, "ms_per_stamp": dt * 1e3 / (n1P * n1P), "stamps_per_block": n1P * n1P,
        "reps": reps, "ms_per_block_all_reps": [r[0] * 1e3 for r in runs], "psf_groups": len(groups), "kernel": cfg.kernel,
        "batches": [len(c) for c in chunks], "input_pixels": int(pool.npool),
        "tables": {"computed": int(tabs.computed_tables), "block_total": int(tabs.block_demand()), "arena": int(tabs.capacity),
                   "evicted": int(tabs.evicted_tables), "spectra_resets": int(tabs.spectra_resets)},
        "workload": f"{config} geometry ({E} exposures, {cfg.kernel}), block of {n1P}x{n1P} output stamps, PSF group per 2x2 InStamps; batch plan + PSF sampling + "
                    "tables + selection + pair maps + A, B, LA kernel, coaddition + block maps inside the timed region; median of the timed blocks",
        "stage_ms_per_block": stages, # (the Eigen kernel's sub-batches run on two streams: their stage times overlap and do not add up to the wall time)
        "host_and_gaps_ms_per_block": (dt * 1e3 - sum(stages.values())) if sum(stages.values()) <= dt * 1e3 else None,
        "out_map_rms": float(maps.out_map.square().mean().sqrt()),
        "batches_run": list(getattr(maps, "chunk_sizes", [])), "info_nonzero": int(getattr(maps, "info_nonzero", 0)),  # the plan of the last timed block itself
        "free_gb_at_end": round(torch.cuda.mem_get_info(dev)[0] / 1e9, 1), "torch_reserved_gb": round(torch.cuda.memory_reserved(dev) / 1e9, 1),
        "passes_halved": int(getattr(maps, "passes_halved", 0)),  # passes the device had no memory for after all, run again as two (last timed block)
        "plan_stable": bool(plan_stable), "memory_plan": {k: mplan[k] for k in ("capacity", "spec_capacity", "stamps", "bytes_per_stamp", "available")},
    }


def config_legs(ctx, dev, which=("cfg1", "cfg4", "cfg5", "cfg3")):
    """The other BASELINE configurations on the same clock as the headline, each with its own roofline:
    compact legs, inputs resident, one warm-up step and a few timed ones.  cfg-1 / 4 / 5 are the Cholesky path (roofline: the
    solve family, algorithmic 2 N^2 m per stamp over the HIP-event time of its launches); cfg-3 is the Eigen path with its
    three-node kappa sweep at batch 32 and 256 (roofline: the path's OWN count 4 N^3 / 3 + 4 N^2 m -- reduction + the two
    applications of the reflectors -- over the time of reduction + applications + search; and the HBM roofline of its one
    bandwidth-bound kernel, symv4: N^3 / 3 bytes of trailing triangle per stamp over the HIP-event time of those launches,
    taken in a second pass because the events sit between 737 dependent launches)."""
    import numpy as np
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    plan = {"cfg1": [(256, 5)], "cfg4": [(256, 3)], "cfg5": [(170, 2)], "cfg3": [(32, 2), (256, 1)]}
    out = {}
    for name in which:
        cfg = synth.CONFIGS[name]
        legs = {}
        for nb, steps in plan[name]:
            stamps = sorted((synth.make_stamp(cfg, i) for i in range(nb)), key=lambda st: -st.n)  # deepest first (cfg-4 is ragged)
            psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
            tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=dev)
            b = StampBatch(cfg, stamps, tables, ctx=ctx, device=dev)
            b.run()
            torch.cuda.synchronize()
            ctx.profile_enable(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                b.run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            fam_names = ("solve_gemm", "chol_gemm", "chol_diag", "build_A", "build_B", "finalize", "eigen_trd", "eigen_applyq", "lakernel1", "eigen_gemm", "epilogue")
            fams = {f: ctx.profile_get(f) for f in fam_names}
            n = b.n.astype(np.float64)
            leg = {"value": nb / dt, "unit": "postage-stamps/s", "ms_per_stamp": dt / nb * 1e3, "ms_per_step": dt * 1e3, "batch": nb, "steps": steps,
                   "N_mean": float(n.mean()), "N_max": int(n.max()), "m": cfg.m, "kernel": cfg.kernel, "kappaC": list(cfg.kappaC),
                   "n_expo": list(cfg.n_expo) if isinstance(cfg.n_expo, tuple) else cfg.n_expo,
                   "stage_ms_per_step": {k: v[0] / steps for k, v in fams.items() if v[0] > 0}, "info_nonzero": int((b.info != 0).sum())}
            if cfg.kernel == "Cholesky":
                flops = float((2.0 * cfg.m * n**2).sum())
                ms, launches = fams["solve_gemm"]
                ach = flops * steps / (ms * 1e-3) / 1e12
                leg["roofline"] = {"kernel": "solve_fwd_kernel+solve_bwd_kernel", "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / FP64_MFMA_PEAK_TFLOPS, "flops_per_launch": flops * steps / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                                   "launches": launches, "traffic": None}
                job = float((n**3 / 3.0 + 2.0 * cfg.m * n**2 + 165.0 * n * (n + 1) + 220.0 * n * cfg.m).sum())
            else:
                flops = float((4.0 * n**3 / 3.0 + 4.0 * n**2 * cfg.m).sum())
                # wall time of the solve: the step minus the builders and the epilogue (the stage events of reduction / reflector products /
                # search overlap in time -- second queue, sub-batches on streams of their own -- so their sum is not a duration)
                ms = dt * 1e3 * steps - sum(fams[k][0] for k in ("build_A", "build_B", "epilogue"))
                ach = flops * steps / (ms * 1e-3) / 1e12
                leg["roofline"] = {"kernel": "Eigen solve: band reduction (symv4 / band_step / band_apply / syr2k) + reflector GEMMs + kappa search", "bound": "mfma",
                                   "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                                   "flops_per_stamp": flops / nb, "solve_wall_ms_per_step": ms / steps, "count": "4 N^3 / 3 + 4 N^2 m (the path's own; SURVEY 8d's 9 N^3 + 4 N^2 m would read "
                                   f"{(9.0 * n**3 + 4.0 * n**2 * cfg.m).sum() * steps / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS:.2f})", "traffic": None}
                job = flops + float((165.0 * n * (n + 1) + 220.0 * n * cfg.m).sum())
                # second pass with HIP events around every symv4 launch (profile level 2), the whole batch on ONE stream (with sub-batches
                # on streams of their own the launches of two streams overlap and share the memory system: durations would not add)
                split_was = os.environ.get("IMCOM_EIGEN_SPLIT")
                os.environ["IMCOM_EIGEN_SPLIT"] = "1"
                ctx.profile_enable(2)
                ctx.profile_reset()
                b.solve()
                torch.cuda.synchronize()
                if split_was is None:
                    os.environ.pop("IMCOM_EIGEN_SPLIT", None)
                else:
                    os.environ["IMCOM_EIGEN_SPLIT"] = split_was
                ms4, l4 = ctx.profile_get("symv4")
                trd2 = ctx.profile_get("eigen_trd")[0]
                if l4:
                    nbytes = float((n**3 / 3.0).sum())  # sum over the groups of four columns of the trailing lower triangle, 8 B per entry
                    per_matrix, src = pmc_band_traffic("symv4_kernel")
                    leg["roofline_hbm"] = {"kernel": "symv4_kernel", "bound": "hbm", "achieved": nbytes / (ms4 * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                           "frac": nbytes / (ms4 * 1e-3) / 8e12, "bytes_per_launch": nbytes / l4, "avg_launch_ms": ms4 / l4, "launches": l4,
                                           "ms_per_step": ms4, "reduction_ms_in_this_pass": trd2,
                                           "traffic": None if per_matrix is None else per_matrix * nb,  # counters: per launch and matrix at N = 2938, x this batch
                                           "traffic_source": src,
                                           "note": "algorithmic bytes N^3/3 per stamp; HIP events around each launch, taken in a separate pass"}
            leg["job_roofline_frac"] = job / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS
            ctx.profile_enable(False)
            legs[f"b{nb}"] = leg
            del b, tables
            torch.cuda.empty_cache()
        out[name] = legs[next(iter(legs))] if len(legs) == 1 else legs
        if len(legs) > 1:  # the headline of a configuration with several batches: the larger one
            big = legs[f"b{max(nb for nb, _ in plan[name])}"]
            out[name] = dict(legs, value=big["value"], unit=big["unit"], ms_per_stamp=big["ms_per_stamp"], roofline=big["roofline"])
        out[name]["config"] = f"BASELINE configs[{CONFIG_INDEX[name]}]"
    return out


def paper4_leg(ctx, dev, batch=128, steps=2, cpu_budget=25.0, block_passes=3):
    """The reference's OWN benchmark shape (configs/paper4_configs/H158_Chol_benchmark.json; synth.CONFIGS["paper4"]): 32 x 32 outputs with
    fade 3 (m = 1444), INPAD 1.24" (rho = 31.7 of n2 = 32), six exposures, six input layers, N ~ 6.2k, kappa / C = 6e-4 -- the regime a
    pyimcom user runs, N / m = 4.3: the factorisation (N^3 / 3 = 81 GFlop) is nearly as large as the solves (2 N^2 m = 112).  With 48-pixel
    PSFs 11 % of A is beyond the overlap tables' reach (psfutil.py:1691-1702 leaves those samples zero): A + kappa I is indefinite and EVERY
    stamp takes _cholesky_wrapper's repair (lakernel.py:262-279), so a step is: A, B, the smallest eigenvalue (api.hip lambda_min_subspace:
    two factorisations at shifts of its own + 128-column solves), the repaired factorisation, the solve, coaddition of six layers -- the
    factorisation of A + kappa I itself, which fails, is attempted by a batch's first step only (StampBatch.run: expect_repair).
    Three figures: (1) resident batches as the headline is measured; (2) the first passes of the production block -- n1P = 84, a PSF group
    per 2 x 2 InStamps, 1849 groups -- through coadd_block with the block planner's own plan, and the seconds per block that rate gives;
    (3) the oracle on the same stamps on the host's cores."""
    import numpy as np
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block, plan_block, release_buffers
    from pyimcom_amd.stamps import BlockTables, PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["paper4"]
    stamps = [synth.make_stamp(cfg, i) for i in range(batch)]
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=dev)
    b = StampBatch(cfg, stamps, tables, ctx=ctx, device=dev)
    b.run()
    torch.cuda.synchronize()
    fam_names = ("solve_gemm", "chol_gemm", "chol_diag", "eigen_repair", "build_A", "build_B", "finalize", "epilogue")
    ctx.profile_enable(True)
    ctx.profile_reset()
    tel = Telemetry(torch.device(dev).index or 0).start()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    telemetry = tel.stop()
    fams = {f: ctx.profile_get(f) for f in fam_names}
    ctx.profile_enable(False)
    n = b.n.astype(np.float64)
    m = cfg.m
    repaired = int((b.info != 0).sum())
    fs_flops = float((n**3 / 3.0 + 2.0 * m * n**2).sum())  # factorisation + solves, what a stamp needs
    fs_ms = sum(fams[k][0] for k in ("solve_gemm", "chol_gemm", "chol_diag", "eigen_repair")) / steps
    ach = fs_flops / (fs_ms * 1e-3) / 1e12
    # the kernels themselves: the two factorisations of a step (the failed one and the repaired one) and the one solve, without the
    # eigenvalue iteration -- the rate the same launches have when no repair is needed
    # factorisations the Cholesky stages timed per step (two launch groups -- update, panel solve -- per block column each): ONE since the
    # steps after the first skip the factorisation that is known to fail (StampBatch.solve_begin(expect_repair=True))
    nbk = (int(n.max()) + 127) // 128
    facts = max(1, int(round(fams["chol_gemm"][1] / steps / (2.0 * nbk))))
    k_ms = (fams["solve_gemm"][0] + (fams["chol_gemm"][0] + fams["chol_diag"][0]) / facts) / steps
    ach_k = fs_flops / (k_ms * 1e-3) / 1e12
    job = fs_flops + float((165.0 * n * (n + 1) + 220.0 * n * m).sum())
    out = {"value": batch / dt, "unit": "postage-stamps/s", "ms_per_stamp": dt / batch * 1e3, "ms_per_step": dt * 1e3, "batch": batch, "steps": steps,
           "N_mean": float(n.mean()), "N_max": int(n.max()), "m": m, "n_inframe": cfg.n_inframe, "n_expo": cfg.n_expo, "fade": cfg.fade, "kappaC": list(cfg.kappaC),
           "stamps_repaired": repaired, "stage_ms_per_step": {k: v[0] / steps for k, v in fams.items() if v[0] > 0},
           "config": "configs/paper4_configs/H158_Chol_benchmark.json of the reference (OUTSIZE [80, 32, 0.0390625], FADE 3, PAD 2, INPAD 1.24, KAPPAC [6e-4], "
                     "NPIXPSF 48, GAUSSIAN target, five EXTRAINPUT layers) at six exposures, analytic Roman-like PSFs",
           "roofline": {"kernel": "factorisation + triangular solves as the path runs them (failed factorisation, smallest-eigenvalue iteration, repaired factorisation, solve)",
                        "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                        "flops_per_stamp": fs_flops / batch, "count": "N^3/3 + 2 N^2 m per stamp (what a stamp needs; the repair's extra factorisations and block solves are "
                        "time, not credit)", "ms_per_step": fs_ms, "traffic": None},
           "roofline_kernels": {"kernel": "chol_update + chol_trsm + chol_diag (one factorisation) + solve_fwd + solve_bwd", "bound": "mfma", "achieved": ach_k,
                                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_k / FP64_MFMA_PEAK_TFLOPS, "ms_per_step": k_ms,
                                "note": "the same launches without the eigenvalue iteration: the rate of a stamp whose A + kappa I is positive definite"},
           "roofline_chol": roofline_chol(n, facts, fams["chol_gemm"][0] / steps, fams["chol_diag"][0] / steps, fams["chol_gemm"][1] // steps),
           "roofline_build_A": roofline_build_A(n, fams["build_A"][0] / steps, samples_on_table=on_table_samples(b, cfg)),
           "job_roofline_frac": job / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS, "telemetry": telemetry}
    cpu_sample = stamps[:4]

    # the kernel-class seam on production stamps (OutStamp.LAKERNEL, coadd.py:839-844): HipCholKernel(outst)() one stamp per call on host
    # arrays (380 MB up, 36 MB down), as the reference's stamp loop calls it -- neighbours one after the other, so the calls after the second
    # expect the repair and start from their predecessors' smallest eigenvalues (lakernel._repair_memory) -- and the four OutStamps of a
    # 2 x 2 group in one call (solve_chol_stamps)
    def kernel_seam():
        from pyimcom_amd.lakernel import HipCholKernel, solve_chol_stamps

        class O:
            pass

        def outst(s):
            n_ = int(b.n[s])
            o, o.blk = O(), O()
            o.blk.cfg = O()
            c = o.blk.cfg
            c.n_out, c.n2f, c.kappaC_arr, c.uctarget, c.sigmamax = 1, cfg.n2f, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax
            o.sysmata = b.A[s, :n_, :n_].cpu().numpy().copy()
            o.mhalfb = np.ascontiguousarray(b.Bt[s, :n_, :m].cpu().numpy().T)[None]
            o.outovlc, o.inpix_cumsum = np.array([tables.C]), np.array([n_])
            return o

        b.build()
        torch.cuda.synchronize()
        outs = [outst(s) for s in range(4)]
        per_call = []
        for o in outs:
            t0 = time.perf_counter()
            HipCholKernel(o, ctx=ctx)()
            per_call.append((time.perf_counter() - t0) * 1e3)
        ref_T = outs[3].T
        group = [outst(s) for s in range(4)]
        t0 = time.perf_counter()
        solve_chol_stamps(group, ctx=ctx)
        t4 = (time.perf_counter() - t0) * 1e3
        return {"ms_per_call": per_call, "ms_per_stamp": per_call[-1], "ms_per_stamp_group4": t4 / 4, "group4_vs_single_T": float(np.abs(group[3].T - ref_T).max() / np.abs(ref_T).max()),
                "what": "HipCholKernel(outst)() on host arrays, one production stamp per call (four neighbours in a row: the last call expects the repair and has "
                        "its predecessors' hint), then lakernel.solve_chol_stamps([four OutStamps]); PCIe-inclusive"}

    try:
        out["kernel_seam"] = kernel_seam()
    except Exception as e:  # noqa: BLE001
        import traceback

        traceback.print_exc(file=sys.stderr)
        out["kernel_seam"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del b, tables
    torch.cuda.synchronize()
    release_buffers()
    ctx.release_workspace()
    torch.cuda.empty_cache()

    # (2) the production block: n1P = 84, a PSF group per 2 x 2 InStamps; the first passes of the planner's plan, timed
    def block():
        n1P = 84
        cfgb, inst, pool, psfs_b, target_b, groups, counts, img_all, yxco_all = block_workload(dev, n1P, config="paper4")
        from pyimcom_amd import psfs as psfmod

        ns, order = psfs_b.shape[-1], {k: q for q, k in enumerate(groups)}

        def sample_groups(keys):
            idx = torch.tensor([order[k] for k in keys]).pin_memory().to(dev, non_blocking=True)
            return psfmod.sample_psf(img_all[idx].reshape(-1, ns + 16, ns + 16), ns, yxco_all[idx].reshape(-1, 2, ns, ns), psf_norm=True, ctx=ctx)

        from pyimcom_amd.blockrun import memory_plan

        mplan = memory_plan(cfgb, pool, n1P, cfgb.n_expo, ctx=ctx)
        tabs = BlockTables(groups, target_b, cfgb.nfft, ctx=ctx, device=dev, group_count=counts, bulk_provider=sample_groups, cells=True,
                           capacity=mplan["capacity"], spec_capacity=mplan["spec_capacity"])
        plan = plan_block(cfgb, pool, tabs, n1P)
        first = plan[:block_passes]
        coadd_block(cfgb, pool, tabs, n1P, cfgb.n_expo, chunks=first[:1], pad_sides=None)  # warm-up: buffers, workspace, kernels
        torch.cuda.synchronize()
        tabs.reset()
        ctx.profile_enable(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        maps = coadd_block(cfgb, pool, tabs, n1P, cfgb.n_expo, chunks=first, pad_sides=None)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        fam_b = ("psf_sample", "psf_spectra", "psf_overlap", "select", "build_A", "build_B", "chol_gemm", "chol_diag", "eigen_repair", "solve_gemm", "finalize", "epilogue", "block_acc")
        st = {f: ctx.profile_get(f)[0] for f in fam_b}
        ctx.profile_enable(False)
        done = sum(len(c) for c in first)
        # a block's first pass attempts the factorisation that fails; the others are told by the pass before them (expect_repair) and skip it:
        # seconds per block = the first pass + the other passes at the rate of the timed ones after the first
        ps = [float(t) for t in getattr(maps, "pass_seconds", [])]
        if len(ps) == len(first) and len(first) > 1:
            rest = sum(ps[1:]) / sum(len(c) for c in first[1:])
            per_block = ps[0] + rest * (n1P * n1P - len(first[0]))
        else:
            rest, per_block = None, dtb / done * n1P * n1P
        return {"n1P": n1P, "pass_seconds": [round(t, 3) for t in ps], "ms_per_stamp_steady": None if rest is None else rest * 1e3, "stamps_per_block": n1P * n1P, "psf_groups": len(groups), "passes_in_plan": len(plan), "pass_sizes": sorted({len(c) for c in plan}),
                "passes_timed": len(first), "stamps_timed": done, "ms_timed": dtb * 1e3, "ms_per_stamp": dtb * 1e3 / done, "value": n1P * n1P / per_block, "value_timed_passes": done / dtb, "unit": "postage-stamps/s",
                "seconds_per_block": per_block, "stamps_repaired": int(getattr(maps, "info_nonzero", 0)),
                "tables": {"computed": int(tabs.computed_tables), "block_total": int(tabs.block_demand()), "arena": int(tabs.capacity)},
                "memory_plan": {k: mplan[k] for k in ("capacity", "spec_capacity", "stamps", "bytes_per_stamp", "available")},
                "stage_ms": {k: v for k, v in st.items() if v > 0}, "input_pixels": int(pool.npool),
                "what": "the first passes of the 84 x 84-stamp production block through coadd_block (plan, PSF sampling, spectra, overlap tables of the passes' "
                        "groups, selection, pair maps, A, B, Cholesky with the repair, coaddition, block maps); seconds_per_block = the first pass + the block's other "
                        "stamps at the rate of the timed passes after the first"}

    try:
        out["block"] = block()
    except Exception as e:  # noqa: BLE001
        import traceback

        traceback.print_exc(file=sys.stderr)
        out["block"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.synchronize()
    release_buffers()
    ctx.release_workspace()
    torch.cuda.empty_cache()
    if cpu_budget and cpu_budget > 0:
        try:
            cb = cpu_baseline(cfg, cpu_sample, psfs, target, cpu_budget, modes=("all",))
            out["cpu_baseline"] = cb
            out["vs_cpu"] = out["value"] / cb["value"]
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def iter_default_leg(ctx, dev, batch=128, steps=2, cpu_budget=20.0, block_n1P=16):
    """The Iterative kernel at the reference's DEFAULT configuration (configs/default_config.json and 70 more of its 152 configs;
    synth.CONFIGS["iter_default"]): 32 x 32 outputs, INPAD 0.6" (rho = 15.36 output pixels), six exposures, six input layers, N ~ 2.8k,
    KAPPAC [0.0], ITERRTOL 1.5e-3, ITERMAX 30 -- lakernel.IterKernel (lakernel.py:533-654) with conjugate_gradient (397-442) per output pixel
    on the ~560 input pixels of its acceptance disc.  A step = A, B, the blocked conjugate gradients (csrc/iter_block.hip: 4 x 4 patches of
    output pixels share the dense sub-matrix of the union of their discs, 16 recurrences per MFMA product), the maps, the clamp of
    coadd.py:1104-1107, the coaddition of six layers.  Three figures: (1) resident batches; (2) a block of ``block_n1P``^2 stamps through
    coadd_block, and next to it the SAME block by the Cholesky kernel at kappa/C = 5e-4 with the rms difference of the coadded science layer
    (the reference's own criterion, tests/pyimcom/test_pyimcom.py:971-978: < 2.5e-3); (3) the oracle on the same stamps on the host.
    Roofline of the CG launches: they stream every patch's (symmetric) sub-matrix once per step -- its lower tiles, 4 up^2 bytes per
    patch and step, flops 2 up^2 x 16: 65 us at the HBM peak against 52 us at the matrix pipe's: HBM is the nearer roof."""
    import dataclasses

    import numpy as np
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block, release_buffers
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["iter_default"]
    stamps = [synth.make_stamp(cfg, i) for i in range(batch)]
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=dev)
    b = StampBatch(cfg, stamps, tables, ctx=ctx, device=dev)
    b.run()
    torch.cuda.synchronize()
    fam_names = ("build_A", "build_B", "iter_gather", "iter_cg", "finalize", "epilogue")
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fams = {f: ctx.profile_get(f) for f in fam_names}
    ctx.profile_enable(False)
    its, steps_px = ctx.iter_stats(batch * cfg.m)
    n = b.n.astype(np.float64)
    cg_ms, cg_launches = fams["iter_cg"]
    tr_cg, tr_src = pmc_kernel_traffic("iter_block_cg_sym_kernel" if its.get("half_storage") else "iter_block_cg_kernel", batch, cfg.name,
                                       "r*_pmc_traffic_iter.json")  # (same batch, same kernel sources: else null)
    cg_s = cg_ms * 1e-3 / steps
    out = {"value": batch / dt, "unit": "postage-stamps/s", "ms_per_stamp": dt / batch * 1e3, "ms_per_step": dt * 1e3, "batch": batch, "steps": steps,
           "N_mean": float(n.mean()), "N_max": int(n.max()), "m": cfg.m, "n_inframe": cfg.n_inframe, "n_expo": cfg.n_expo, "kappaC": list(cfg.kappaC),
           "iter_rtol": cfg.iter_rtol, "iter_max": cfg.iter_max, "rho_acc": cfg.rho, "kernel": cfg.kernel,
           "stage_ms_per_step": {k: v[0] / steps for k, v in fams.items() if v[0] > 0},
           "patches": its["patches"], "max_union": its["max_union"], "blocked_solver": its["blocked"],
           "cg_steps_mean": float(steps_px.mean()), "cg_steps_max": int(steps_px.max()), "cg_pixels_at_itermax": float((steps_px >= cfg.iter_max).mean()),
           "cg_steps_per_patch": its["patch_steps"] / max(its["patches"], 1),
           "config": "configs/default_config.json of the reference (LAKERNEL Iterative, KAPPAC [0.0], OUTSIZE [80, 32, 0.0390625], INPAD 0.6, ITERRTOL 1.5e-3, "
                     "ITERMAX 30, NPIXPSF 48, GAUSSIAN target, five EXTRAINPUT layers) at six exposures, analytic Roman-like PSFs",
           "roofline": {"kernel": ("iter_block_cg_sym_kernel" if its.get("half_storage") else "iter_block_cg_kernel") + " (16 conjugate-gradient recurrences per 4 x 4 "
                        "patch, one MFMA product per step)", "bound": "hbm",
                        "achieved": its["bytes"] / cg_s / 1e9 if cg_s else None, "peak": 8000.0, "unit": "GB/s", "frac": its["bytes"] / cg_s / 8e12 if cg_s else None,
                        "bytes_per_launch": its["bytes"] * steps / max(cg_launches, 1), "avg_launch_ms": cg_ms / max(cg_launches, 1), "launches": cg_launches // max(steps, 1),
                        "mfma_TFLOPs": its["flops"] / cg_s / 1e12 if cg_s else None, "mfma_frac": its["flops"] / cg_s / 1e12 / FP64_MFMA_PEAK_TFLOPS if cg_s else None,
                        "half_storage": bool(its.get("half_storage")),
                        "count": "per patch and CG step: the 16 x 16 tiles on and below the diagonal of the union's sub-matrix = (nt (nt + 1) / 2) x 2 KB (nt = union "
                                 "/ 16 rounded up; the full-storage kernel of unions above 864 rows reads both triangles: 8 up^2), 32 up^2 flops",
                        "traffic": tr_cg, "traffic_source": tr_src}}
    job = its["flops"] + float((165.0 * n * (n + 1) + 220.0 * n * cfg.m).sum())
    out["job_roofline_frac"] = job / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS
    cpu_sample = stamps[:8]
    del b, tables
    torch.cuda.synchronize()
    release_buffers()
    ctx.release_workspace()
    torch.cuda.empty_cache()

    def block():
        n1P, E = block_n1P, cfg.n_expo
        rng = np.random.default_rng(11)
        inst = synth.make_instamps(cfg, n1P, E, rng)
        # the science layer: a field of unit-flux stars seen through a Gaussian of 0.9 native pixels (smooth, like the reference test's image)
        p = synth.NATIVE_ARCSEC / cfg.dtheta_as
        span = n1P * cfg.n2
        sx, sy = rng.uniform(0, span, 24), rng.uniform(0, span, 24)
        sig = 0.9 * p
        inst2 = []
        for (px, py, data, cum) in inst:
            d = data.copy()
            d[0] = (np.exp(-0.5 * ((px[:, None] - sx[None]) ** 2 + (py[:, None] - sy[None]) ** 2) / sig**2).sum(axis=1) / (2 * np.pi * 0.9**2)).astype(np.float32)
            inst2.append((px, py, d, cum))
        pool = InStampPool(inst2, cfg.n_inframe, device=dev)
        tabs = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=dev)
        coadd_block(cfg, pool, tabs, n1P, E, pad_sides=None)  # warm-up (buffers, workspace)
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        maps = coadd_block(cfg, pool, tabs, n1P, E, pad_sides=None)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        st = {f: ctx.profile_get(f)[0] for f in ("select",) + fam_names}
        ctx.profile_enable(False)
        img_it = maps.out_map[0, 0].cpu().numpy().copy()
        passes = list(getattr(maps, "chunk_sizes", []))
        del maps
        release_buffers()
        ctx.release_workspace()
        torch.cuda.empty_cache()
        # the same block by the Cholesky kernel at the kappa the reference's own end-to-end test uses for this comparison (5e-4 C)
        chol = dataclasses.replace(cfg, kernel="Cholesky", kappaC=(5e-4,))
        t0 = time.perf_counter()
        mc = coadd_block(chol, pool, tabs, n1P, E, pad_sides=None)
        torch.cuda.synchronize()
        dtc = time.perf_counter() - t0
        img_ch = mc.out_map[0, 0].cpu().numpy().copy()
        lo, hi = cfg.n2 // 2, n1P * cfg.n2 - cfg.n2 // 2  # (the block's inner part: the boundary recovery is left out on both sides)
        d = (img_it - img_ch)[lo:hi, lo:hi]
        return {"n1P": n1P, "stamps_per_block": n1P * n1P, "value": n1P * n1P / dtb, "unit": "postage-stamps/s", "ms_per_stamp": dtb * 1e3 / (n1P * n1P), "ms_per_block": dtb * 1e3,
                "passes": passes, "stage_ms": {k: v for k, v in st.items() if v > 0}, "cholesky_block_ms": dtc * 1e3,
                "seconds_per_production_block": 84 * 84 * dtb / (n1P * n1P),  # OUTSIZE [80, 32, .] with PAD 2: 84 x 84 stamps, at this block's rate
                "image_rms_vs_cholesky": float(d.std()), "image_mean_vs_cholesky": float(d.mean()), "image_std": float(img_ch[lo:hi, lo:hi].std()),
                "image_peak": float(np.abs(img_ch).max()), "criterion": "std < 2.5e-3 and |mean| < 2e-4 (tests/pyimcom/test_pyimcom.py:971-978)",
                "what": "a block of n1P^2 stamps through coadd_block (ONE PSF group; selection, A, B, CG, coaddition, block maps), then the same block "
                        "by the Cholesky kernel at kappa/C = 5e-4; rms of the science layer's difference"}

    try:
        out["block"] = block()
        out["image_rms_vs_cholesky"] = out["block"]["image_rms_vs_cholesky"]
    except Exception as e:  # noqa: BLE001
        import traceback

        traceback.print_exc(file=sys.stderr)
        out["block"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.synchronize()
    release_buffers()
    ctx.release_workspace()
    torch.cuda.empty_cache()
    if cpu_budget and cpu_budget > 0:
        try:
            cb = cpu_baseline(cfg, cpu_sample, psfs, target, cpu_budget, modes=("all",))
            out["cpu_baseline"] = cb
            out["vs_cpu"] = out["value"] / cb["value"]
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def summary_of(out):
    """Every leg of the line in a few numbers -- value and roofline fraction -- short enough to survive in the last ~1.5 k characters of
    the line (a driver that keeps a tail of stdout sees all legs, not only the last ones).  v = stamps/s, ms = ms per stamp, f = roofline
    fraction of the leg's dominant kernel family, job = all algorithmic flops of the stamps over the wall time / fp64 matrix peak."""
    r3 = lambda x: None if x is None else float(f"{x:.4g}")  # noqa: E731

    def leg(d, extra=()):
        if not isinstance(d, dict):
            return None
        if "error" in d:
            return {"error": d["error"][:60]}
        o = {"v": r3(d.get("value")), "ms": r3(d.get("ms_per_stamp"))}
        if isinstance(d.get("roofline"), dict):
            o["f"] = r3(d["roofline"].get("frac"))
        if d.get("job_roofline_frac") is not None:
            o["job"] = r3(d["job_roofline_frac"])
        for k in extra:
            if d.get(k) is not None:
                o[k] = r3(d[k]) if isinstance(d[k], float) else d[k]
        return o

    sm = {"headline": {"v": r3(out["value"]), "ms_step": r3(out["ms_per_step"]), "f": r3(out["roofline"]["frac"]), "launch_ms": r3(out["roofline"]["avg_launch_ms"]),
                       "step_ms": [r3(out["step_ms"][k]) for k in ("min", "median", "max")] if out.get("step_ms", {}).get("min") is not None else None,
                       "probe": [r3(out["roofline"].get("mfma_probe_tflops")), r3(out["roofline"].get("mfma_probe_sustained_tflops"))],
                       "n_gpus": out["n_gpus"]}}
    tel = out.get("telemetry") or {}
    if tel.get("sclk_mhz"):
        sm["headline"]["sclk"] = [r3(tel["sclk_mhz"][k]) for k in ("min", "median", "max")]
        sm["headline"]["W"] = r3(tel.get("power_w", {}).get("median"))
        sm["headline"]["Tj"] = r3(tel.get("temp_junction_c", {}).get("max"))
    for k in ("roofline_chol", "roofline_build_A"):
        if isinstance(out.get(k), dict):
            sm["headline"][k[9:]] = r3(out[k]["frac"])
    if "block" in out:
        sm["block"] = leg(out["block"])
    if "eigen_block" in out:
        sm["eigen_block"] = leg(out["eigen_block"], ("batches_run",))
    if "block_seam" in out:
        bs = out["block_seam"]
        sm["block_seam"] = {"v": r3(bs.get("value")), "threads": bs.get("host_threads"), "v1": r3(bs.get("one_host_thread", {}).get("value")),
                            "v1_exact": r3(bs.get("one_host_thread_exact_positions", {}).get("value")), "v_passes": r3(bs.get("several_passes", {}).get("value"))}
    if "kernel_seam" in out:
        ks = out["kernel_seam"]
        sm["kernel_seam"] = {"ms": r3(ks.get("ms_per_stamp")), "ms4": r3(ks.get("ms_per_stamp_group4"))}
    cf = out.get("configs")
    if isinstance(cf, dict):
        if "error" in cf:
            sm["configs"] = {"error": cf["error"][:60]}
        for name in ("cfg1", "cfg4", "cfg5"):
            if name in cf:
                sm[name] = leg(cf[name])
        if "cfg3" in cf:
            c3 = cf["cfg3"]
            sm["cfg3"] = {k: {"ms": r3(v.get("ms_per_stamp")), "f": r3(v.get("roofline", {}).get("frac")), "symv4_f": r3(v.get("roofline_hbm", {}).get("frac"))}
                          for k, v in c3.items() if isinstance(v, dict) and k.startswith("b")} if isinstance(c3, dict) and "error" not in c3 else leg(c3)
        if "paper4" in cf:
            p4 = cf["paper4"]
            sm["paper4"] = leg(p4, ("stamps_repaired", "batch"))
            if isinstance(p4, dict) and "error" not in p4:
                sm["paper4"]["f_kernels"] = r3(p4.get("roofline_kernels", {}).get("frac"))
                blk = p4.get("block", {})
                sm["paper4"]["block"] = {"error": blk["error"][:60]} if "error" in blk else {"v": r3(blk.get("value")), "s_per_block": r3(blk.get("seconds_per_block"))}
                if isinstance(p4.get("cpu_baseline"), dict) and "value" in p4["cpu_baseline"]:
                    sm["paper4"]["cpu"] = r3(p4["cpu_baseline"]["value"])
                ks4 = p4.get("kernel_seam", {})
                sm["paper4"]["seam_ms"] = {"error": ks4["error"][:60]} if "error" in ks4 else [r3(ks4.get("ms_per_stamp")), r3(ks4.get("ms_per_stamp_group4"))]
        if "iter_default" in cf:
            it_ = cf["iter_default"]
            sm["iter_default"] = leg(it_, ("cg_steps_mean", "image_rms_vs_cholesky"))
            if isinstance(it_, dict) and "error" not in it_:
                if isinstance(it_.get("roofline"), dict):
                    sm["iter_default"]["bound"] = it_["roofline"].get("bound")
                if isinstance(it_.get("block"), dict):
                    sm["iter_default"]["block_v"] = {"error": it_["block"]["error"][:60]} if "error" in it_["block"] else r3(it_["block"].get("value"))
                    sm["iter_default"]["s_per_block"] = r3(it_["block"].get("seconds_per_production_block"))
                if isinstance(it_.get("cpu_baseline"), dict) and "value" in it_["cpu_baseline"]:
                    sm["iter_default"]["cpu"] = r3(it_["cpu_baseline"]["value"])
    if "farm" in out:
        sm["farm"] = leg(out["farm"], ("makespan_s", "blocks", "ranks_seen"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        sm["cpu"] = {"v": r3(cb["value"]), "cores": cb.get("cores"), "v1": r3(cb.get("one_thread", {}).get("value")), "vP": r3(cb.get("processes", {}).get("value")),
                     "P": cb.get("processes", {}).get("processes")}
    return sm


def _sig(x, digits=5):
    """Floats of the compact line carry `digits` significant digits (the detail file keeps everything)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


COMPACT_LIMIT = 4000  # bytes of the line a driver parses (a 22 KB line came back as parsed = null)


def compact_line(out, detail_path=None, limit=COMPACT_LIMIT):
    """The ONE line bench.py prints on stdout, last: the contract's keys, the dominant kernel's roofline, the CPU baseline and the
    summary of every leg, below ``limit`` bytes whatever the legs reported (every verbose object -- configs, block, eigen_block,
    telemetry, per-step times, stage splits, the legs' own rooflines -- is in the detail file and on stderr).  Should the line still
    exceed the limit (it never has: ~2.6 KB with every leg), summary entries are dropped from the end until it fits; the contract's
    keys, ``roofline`` and ``cpu_baseline`` are never dropped."""
    keep = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "ms_per_stamp", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cf = out.get("config", {})
    line["config"] = {k: (cf[k][:200] if isinstance(cf[k], str) else cf[k]) for k in ("workload", "stamps_per_step_per_gpu", "N_mean", "m", "parallelism") if k in cf}
    rf = out.get("roofline", {})
    line["roofline"] = {k: (rf[k][:120] if isinstance(rf[k], str) else rf[k]) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                                                         "avg_launch_ms", "launches", "flops_per_launch") if k in rf}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        if "error" in cb:
            line["cpu_baseline"] = {"error": cb["error"][:120]}
        else:
            c = {k: (cb[k][:160] if isinstance(cb[k], str) else cb[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
            if isinstance(cb.get("one_thread"), dict):
                c["one_thread"] = cb["one_thread"].get("value")
            if isinstance(cb.get("processes"), dict) and "value" in cb["processes"]:
                c["processes"] = {"value": cb["processes"]["value"], "processes": cb["processes"].get("processes")}
            line["cpu_baseline"] = c
    if out.get("farm") is not None and isinstance(out["farm"], dict):
        line["farm"] = {k: out["farm"].get(k) for k in ("value", "makespan_s", "blocks", "ranks_seen", "host_build_s", "per_rank_busy_s", "error") if out["farm"].get(k) is not None}
        if isinstance(line["farm"].get("error"), str):
            line["farm"]["error"] = line["farm"]["error"][:120]
    if out.get("per_rank_value") and out.get("n_gpus", 1) > 1:
        line["per_rank_value"] = out["per_rank_value"]
    line["summary"] = dict(out.get("summary") or {})
    line["detail"] = detail_path
    line = {k: (_sig(v, 10) if isinstance(v, float) else _sig(v)) for k, v in line.items()}  # (the contract's own numbers keep their digits: value x ms_per_step is checked)
    text = json.dumps(line, separators=(",", ":"))
    while len(text) > limit and line["summary"]:
        line["summary"].pop(next(reversed(line["summary"])))
        line["summary_truncated"] = True
        text = json.dumps(line, separators=(",", ":"))
    for k in ("farm", "per_rank_value"):  # (an 8-rank farm leg's lists: still not at the price of the contract's keys)
        if len(text) > limit and k in line:
            line.pop(k)
            text = json.dumps(line, separators=(",", ":"))
    return text


def write_detail(out):
    """Everything the legs reported, as JSON, beside bench.py (``bench_detail.json``; also under gpurun_out/ when that directory exists,
    so that a gpurun call brings it back) -- and on stderr, one "[bench detail] key: value" line per top-level key."""
    text = json.dumps(out)
    paths = [os.environ.get("IMCOM_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not os.environ.get("IMCOM_BENCH_DETAIL"):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                f.write(text + "\n")
            written = written or os.path.relpath(p_, ROOT)
        except OSError:
            pass
    for k, v in out.items():  # stderr: one prefixed line per top-level key (never a bare JSON line: stdout's last line is the only one of those)
        print(f"[bench detail] {k}: {json.dumps(v)}", file=sys.stderr)
    sys.stderr.flush()
    return written


def farm_leg(rank, world, dev, dist, cdev, mosaic=4, n1P=32, config="cfg4", seed=4):
    """BASELINE configs[3] on the multi-GPU clock: a mosaic of ``mosaic`` x ``mosaic`` blocks (every block its own exposure depth in 6-10,
    its own lattices, data and PSFs) farmed over the ranks by ``pyimcom_amd.farm.run`` with the dynamic schedule -- blocks claimed on
    start, largest first; the passes of the last blocks shared; no data-path collective, coordination through files of the output
    directory, as the reference's own one-process-per-block runs coordinate through output files (docs/run_README.rst:81-100,
    examples/multiblock_norep.pl:25-27, 42-66).  Inside the clock (barrier to barrier): every block's inputs (make_block: InStamps, sampled
    PSFs, overlap tables), the plan, the passes, the merges of shared blocks, the block files.  Collectives are used for the clock and the
    accounting only.  Every rank calls this; rank 0 gets the dict."""
    import shutil
    import tempfile

    import numpy as np
    import torch

    from pyimcom_amd import farm

    blocks, costs, make_block = farm.synthetic_mosaic(config, mosaic, n1P, seed)
    box = [tempfile.mkdtemp(prefix="imcom_farm_") if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    outdir = box[0]
    # every block's host inputs (InStamp lattices and data, sampled analytic PSFs: synthesis, the stand-in for a survey's files) are built
    # BEFORE the clock, by every rank (the dynamic schedule decides at run time who takes which block): the leg times GPUs, not the
    # synthesis (it was 13 of a 37 s makespan).  What stays inside: upload, tables, plan, passes, merges, files.
    t_h = time.perf_counter()
    host_cache = {b: make_block.host(b) for b in blocks}
    host_build_s = time.perf_counter() - t_h
    mk = lambda b: make_block.device(b, host_cache[b], dev)  # noqa: E731
    mk.host, mk.device = (lambda b: host_cache[b]), (lambda b, h: make_block.device(b, h, dev))
    stats, err = {}, None
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    try:
        farm.run(blocks, costs, mk, outdir, rank, world, device=dev, restart=False, log=lambda *_: None, stats=stats, max_wait=600.0)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001  (a rank that fails still meets the others in the collectives below, then exits non-zero)
        import traceback

        traceback.print_exc(file=sys.stderr)
        err = f"{type(e).__name__}: {e}"[:300]
    mine = time.perf_counter() - t0
    vec = torch.zeros(4 * world + 2, dtype=torch.float64, device=cdev)
    vec[rank], vec[world + rank], vec[2 * world + rank], vec[3 * world + rank] = mine, stats.get("busy_s", 0.0), stats.get("blocks_written", 0), stats.get("passes_run") or 0
    vec[4 * world], vec[4 * world + 1] = (1.0 if err else 0.0), 1.0
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    vec = vec.cpu().numpy()
    failed, ranks_seen = int(round(vec[4 * world])), int(round(vec[4 * world + 1]))
    out = None
    if rank == 0:
        wall = vec[:world]
        stamps = len(blocks) * n1P * n1P
        missing = [b for b in blocks if not os.path.exists(farm.block_path(outdir, b))]
        ok = not failed and not missing
        if ok:
            z = np.load(farm.block_path(outdir, blocks[0]))
            ok = bool(np.isfinite(z["out_map"]).all() and np.abs(z["out_map"]).max() > 0)
        out = {"value": stamps / float(wall.max()) if ok else None, "unit": "postage-stamps/s", "makespan_s": float(wall.max()), "blocks": len(blocks),
               "stamps": stamps, "stamps_per_block": n1P * n1P, "ranks_seen": ranks_seen, "n_gpus": world,
               "per_rank_wall_s": [float(v) for v in wall], "per_rank_busy_s": [float(v) for v in vec[world : 2 * world]],
               "per_rank_blocks_written": [int(round(v)) for v in vec[2 * world : 3 * world]], "per_rank_passes": [int(round(v)) for v in vec[3 * world : 4 * world]],
               "ms_per_stamp": float(wall.max()) / stamps * 1e3 * world if ok else None,  # GPU-milliseconds per stamp
               "host_build_s": host_build_s,  # rank 0's synthesis of all blocks' host inputs, before the clock
               "busy_share": float(np.mean(vec[world : 2 * world]) / wall.max()) if wall.max() > 0 else None,
               "schedule": "dynamic (blocks claimed largest first, tail passes shared)", "config": f"BASELINE configs[3]: {mosaic}x{mosaic} blocks of {n1P}x{n1P} "
               f"{config} stamps, 6-10 exposures per block; upload + tables + plan + passes + merges + block files inside the clock (host inputs synthesised before it)",
               "ranks_failed": failed, "blocks_missing": missing}
        if not ok:
            out["error"] = err or f"{failed} rank(s) failed, blocks missing: {missing}"
    if dist is not None:
        dist.barrier()
    if rank == 0:
        shutil.rmtree(outdir, ignore_errors=True)
    if failed:  # every rank knows (the all-reduced count): all leave together, non-zero, instead of rank 0 waiting in a later barrier
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(f"[bench rank {rank}] farm leg failed on {failed} rank(s)" + (f": {err}" if err else ""))
    return out


def seam_legs(ctx, dev, cfg, batch):
    """The two seams a pyimcom user reaches the library through, timed the way they are used (never the headline):
    ``kernel_seam`` -- the drop-in LA kernel class (OutStamp.LAKERNEL, coadd.py:839-844, 1091-1093): ONE stamp per call, A and -B/2
    handed over as host arrays (100 MB over PCIe), T and the maps returned as host arrays; ``block_seam`` -- the reference's
    Block containers through ``refblock.coadd_output_stamps`` (coadd.py:2003-2084): a duck-typed 16 x 16-stamp block at cfg-2
    geometry, PSF images fetched per 2 x 2 group and exposure from the host-side objects (get_psf_pos, outpix2world2inpix on
    nsamp^2 positions each: the host share is the reference's own WCS / file-broker work, here an affine map), block maps
    returned as host arrays."""
    import numpy as np
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.lakernel import HipCholKernel
    from pyimcom_amd.refblock import coadd_output_stamps

    out = {}
    # kernel-class seam on the first stamp of the headline batch
    n, m = int(batch.n[0]), cfg.m
    A = batch.A[0, :n, :n].cpu().numpy().copy()
    mB = np.ascontiguousarray(batch.Bt[0, :n, :m].cpu().numpy().T)[None]

    class O:
        pass

    def outst():
        o, o.blk = O(), O()
        o.blk.cfg = O()
        c = o.blk.cfg
        c.n_out, c.n2f, c.kappaC_arr, c.uctarget, c.sigmamax = 1, cfg.n2f, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax
        o.sysmata, o.mhalfb, o.outovlc, o.inpix_cumsum = A, mB, np.array([batch.tables.C]), np.array([n])
        return o

    best = None
    for _ in range(4):
        o = outst()
        t0 = time.perf_counter()
        HipCholKernel(o, ctx=ctx)()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["kernel_seam"] = {"ms_per_stamp": best * 1e3, "value": 1.0 / best, "unit": "postage-stamps/s", "N": n, "m": m,
                          "what": "HipCholKernel(outst)() on host arrays, one stamp per call, PCIe-inclusive (best of 4)"}
    # the same seam with the four OutStamps of a 2 x 2 group handed over in one call (each with its own host arrays)
    from pyimcom_amd.lakernel import solve_chol_stamps

    best4, single = None, None
    for _ in range(4):
        group = [outst() for _ in range(4)]
        t0 = time.perf_counter()
        solve_chol_stamps(group, ctx=ctx)
        dt = time.perf_counter() - t0
        best4 = dt if best4 is None else min(best4, dt)
        single = group[0]
    dT = float(np.abs(single.T - o.T).max() / np.abs(o.T).max())
    out["kernel_seam"].update({"ms_per_stamp_group4": best4 * 1e3 / 4, "value_group4": 4.0 / best4, "group4_vs_single_T": dT,
                               "what_group4": "lakernel.solve_chol_stamps([four OutStamps]) on host arrays, one batched factorisation / solve per "
                                              "call, PCIe-inclusive (best of 4)"})
    # Block seam
    n1P = 16
    blk, psfgrp, _, _ = synth.duck_block(cfg, n1P, cfg.n_expo if isinstance(cfg.n_expo, int) else cfg.n_expo[1], seed=5)
    fams = ("psf_sample", "psf_spectra", "psf_overlap", "select", "build_A", "build_B", "chol_gemm", "chol_diag", "solve_gemm", "finalize", "epilogue", "block_acc")
    dctx = ctx
    threads = int(os.environ.get("IMCOM_BENCH_HOST_THREADS", max(1, min(16, (os.cpu_count() or 2) // 2))))
    coadd_output_stamps(blk, psfgrp, device=dev, ctx=ctx, host_threads=threads)
    torch.cuda.synchronize()
    res = {}
    for nthr in (threads, 1, "exact"):  # the host half of the PSF groups on worker threads ahead of the device; on ONE worker (the safe default);
        # and on one worker with every sampling position evaluated on the host as the reference does (positions="exact": round 5's seam)
        dctx.profile_enable(True)
        dctx.profile_reset()
        t0 = time.perf_counter()
        maps = coadd_output_stamps(blk, psfgrp, device=dev, ctx=ctx, host_threads=1 if nthr == "exact" else nthr, positions="exact" if nthr == "exact" else "lattice")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gpu_ms = sum(dctx.profile_get(f)[0] for f in fams)
        dctx.profile_enable(False)
        res[nthr] = (dt, gpu_ms)
        del maps
    dt, gpu_ms = res[threads]
    out["block_seam"] = {"value": n1P * n1P / dt, "unit": "postage-stamps/s", "ms_per_block": dt * 1e3, "stamps_per_block": n1P * n1P,
                         "gpu_ms_per_block": gpu_ms, "host_share": 1.0 - gpu_ms * 1e-3 / dt, "host_threads": threads,
                         "one_host_thread": {"value": n1P * n1P / res[1][0], "ms_per_block": res[1][0] * 1e3, "host_share": 1.0 - res[1][1] * 1e-3 / res[1][0]},
                         "one_host_thread_exact_positions": {"value": n1P * n1P / res["exact"][0], "ms_per_block": res["exact"][0] * 1e3},
                         "positions": "lattice (17 x 17 WCS evaluations per PSF group and exposure, the device forms the 383^2 sampling positions)",
                         "what": "refblock.coadd_output_stamps(blk, PSFGrp) on a duck-typed 16x16-stamp Block (cfg-2 geometry, 81 PSF groups): pool upload, "
                                 "PSF images + sampling positions per group from host objects (prepared on worker threads ahead of the device), everything "
                                 "else on the device, maps back to host",
                         "out_map_rms": float(np.sqrt(np.mean(np.square(blk.out_map))))}
    # the same seam on a block of several passes (32 x 32 stamps, 289 PSF groups): the host half of pass k + 1 is prepared while pass k is on
    # the device, which a one-pass block cannot show (its device waits for the host's whole share first)
    del blk, psfgrp
    n1P = int(os.environ.get("IMCOM_BENCH_SEAM_N1P", "32"))
    if n1P > 16:
        blk, psfgrp, _, _ = synth.duck_block(cfg, n1P, cfg.n_expo if isinstance(cfg.n_expo, int) else cfg.n_expo[1], seed=5)
        coadd_output_stamps(blk, psfgrp, device=dev, ctx=ctx, host_threads=threads)  # (a first block of a size pays its allocations: 2.5 s against 1.1)
        torch.cuda.synchronize()
        dctx.profile_enable(True)
        dctx.profile_reset()
        t0 = time.perf_counter()
        maps = coadd_output_stamps(blk, psfgrp, device=dev, ctx=ctx, host_threads=threads)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gpu_ms = sum(dctx.profile_get(f)[0] for f in fams)
        dctx.profile_enable(False)
        out["block_seam"]["several_passes"] = {"value": n1P * n1P / dt, "unit": "postage-stamps/s", "ms_per_block": dt * 1e3, "stamps_per_block": n1P * n1P,
                                               "gpu_ms_per_block": gpu_ms, "host_share": 1.0 - gpu_ms * 1e-3 / dt, "host_threads": threads,
                                               "what": f"the same call on a {n1P}x{n1P}-stamp Block ({(n1P // 2 + 1) ** 2} PSF groups; the second block of that size in the process)"}
        del maps, blk, psfgrp
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batch", type=int, default=256, help="stamps per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-block", action="store_true", help="skip the block-level leg (tables + selection + stamps + block maps)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE configurations (cfg-1, 3, 4, 5; Eigen block)")
    ap.add_argument("--block-reps", type=int, default=3, help="timed blocks of the block leg (the median is reported)")
    ap.add_argument("--cpu-worker", type=float, default=None, help=argparse.SUPPRESS)  # one single-threaded oracle process (cpu_baseline's P-process figure)
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--ldn", type=int, default=None, help="leading dimension of A / L / -B/2 (experiments; default: N rounded up to 128)")
    ap.add_argument("--farm", action="store_true", help="run the mosaic farm leg (BASELINE configs[3]) also at --gpus 1; at --gpus N > 1 it always runs")
    ap.add_argument("--no-farm", action="store_true", help="skip the farm leg at --gpus N > 1")
    ap.add_argument("--farm-mosaic", type=int, default=4, help="blocks per side of the farm leg's mosaic")
    ap.add_argument("--farm-n1P", type=int, default=32, help="output stamps per block side in the farm leg")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="rehearsal of the multi-rank path on a box with ONE GPU: every rank uses cuda:0 and the ranks "
                         "rendezvous over gloo (not a measurement)")
    args = ap.parse_args()
    if args.cpu_worker is not None:
        return cpu_worker(args.config, args.cpu_worker)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.rehearse_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    from pyimcom_amd import synth
    from pyimcom_amd._lib import Context
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS[args.config]
    dev = f"cuda:{local_rank}"
    ctx = Context(local_rank)
    # each rank coadds its own stamps (block farming: stamp ids are disjoint across ranks)
    stamps = [synth.make_stamp(cfg, rank * args.batch + i) for i in range(args.batch)]
    cpu_sample = stamps[:64]  # the CPU baseline's sample: the first stamps as generated, not the deepest
    if os.environ.get("IMCOM_BENCH_SORT", "1") != "0":
        # A block driver visits its stamps in the order it likes: deepest first.  The solve places stamp s on XCD s mod 8, and the
        # late block rows of a ragged batch (cfg-4) are then shared evenly by the eight XCDs instead of binomially.
        stamps.sort(key=lambda st: -st.n)
    n_expo = max(s.n_expo for s in stamps)
    psfs, target = synth.make_psfs(cfg, n_expo)
    tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=dev)
    batch = StampBatch(cfg, stamps, tables, ctx=ctx, device=dev, ldn=args.ldn)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    barrier()
    ctx.profile_enable(True)
    ctx.profile_reset()
    tel = Telemetry(local_rank).start()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # per-step times: events on the stream the steps run on
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        batch.run()
        marks[k + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    telemetry = tel.stop()
    step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    fams = {f: ctx.profile_get(f) for f in ("solve_gemm", "solve_dinv", "chol_gemm", "chol_diag", "eigen_repair", "build_A", "build_B",
                                            "finalize", "epilogue")}
    ctx.profile_enable(False)
    n_keep, info_keep = batch.n.copy(), batch.info.copy()
    ranks_seen, per_rank = 1, [args.batch * args.steps / elapsed]
    if dist is not None:
        cdev = "cpu" if args.rehearse_shared_gpu else dev
        mine = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # proof that the collective library saw N ranks (the JSON's n_gpus is not just WORLD_SIZE from the environment), and every
        # rank's own rate (its own clock around the same K steps)
        one = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
        rates = torch.zeros(world, dtype=torch.float64, device=cdev)
        rates[rank] = args.batch * args.steps / mine
        dist.all_reduce(rates, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in rates.cpu()]
        if ranks_seen != world:
            raise SystemExit(f"the all-reduce saw {ranks_seen} ranks, WORLD_SIZE says {world}")

    farm_out = None
    if world > 1 and not args.no_farm:
        # every rank: the mosaic of BASELINE configs[3] farmed over the ranks (the headline above is N copies of one batch; this is the
        # configuration BASELINE names for N GPUs)
        from pyimcom_amd.blockrun import release_buffers as _rb

        batch = None
        torch.cuda.synchronize()
        ctx.release_workspace()
        torch.cuda.empty_cache()
        cdev_ = "cpu" if (args.rehearse_shared_gpu or dist is None) else dev
        farm_out = farm_leg(rank, world, dev, dist, cdev_, mosaic=args.farm_mosaic, n1P=args.farm_n1P)
        _rb()
        batch = None
    if rank == 0:
        n_arr = n_keep.astype(np.float64)
        # Algorithmic flops of the launches being timed.  SURVEY 8d counts 2 N^2 m per stamp for the two triangular
        # solves, and the solve_fwd / solve_bwd launches carry all of it: the updates below the 128-row diagonal
        # blocks and the diagonal blocks themselves (priced as the triangular solves they are, m rows_k^2 each,
        # although the kernel multiplies by the dense inverse).  With IMCOM_SOLVE_UNFUSED=1 the diagonal blocks are
        # separate solve_dinv launches, timed apart, and only 2 m (N^2 - sum_k rows_k^2) is attributed here.
        rows_sq = np.array([(np.minimum(128, np.maximum(n - 128 * np.arange((n + 127) // 128), 0)) ** 2).sum() for n in n_keep], dtype=np.float64)
        fused = fams["solve_dinv"][1] == 0
        solve_flops_step = float((2.0 * cfg.m * (n_arr**2 - (0.0 if fused else 1.0) * rows_sq)).sum())
        ms, launches = fams["solve_gemm"]
        traffic, traffic_src = pmc_traffic(args.batch, cfg.name)
        achieved = solve_flops_step * args.steps / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # a pure fp64 MFMA loop on every SIMD, measured now: what the pipe gives with no operand traffic -- for 50 ms (the clock of a
        # cold chip) and sustained for more than a second (the clock the chip holds once it is warm and at its power limit)
        probe = ctx.mfma_probe(50.0)
        tel_p = Telemetry(local_rank).start()
        probe_sustained = ctx.mfma_probe(1200.0)
        probe_tel = tel_p.stop()
        out = {
            "metric": "postage-stamps/sec (and ms/stamp) for N~2k A-solve",
            "value": world * args.batch * args.steps / elapsed,
            "unit": "postage-stamps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,  # all_reduce(SUM) of ones over the process group
            "per_rank_value": per_rank,  # each rank's own stamps/s over its own clock; `value` uses the slowest rank's time
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_stamp": elapsed / args.steps / args.batch * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[{CONFIG_INDEX.get(cfg.name, '?')}] ({cfg.name}): one batch of {args.batch} {cfg.n2}x{cfg.n2}-output stamps "
                            f"(fade {cfg.fade}) per GPU per step, {cfg.n_expo} exposures, "
                            f"{'Gaussian' if cfg.psf == 'gauss' else 'analytic Roman-like'} PSF, Cholesky kappa/C={cfg.kappaC[0]:g}, fp64",
                "stamps_per_step_per_gpu": args.batch,
                "N_mean": float(n_arr.mean()),
                "m": cfg.m,
                "parallelism": f"block-farming x{world} (no collective)",
            },
            "roofline": {
                "kernel": "solve_fwd_kernel+solve_bwd_kernel (blocked triangular solves: block-row updates" + (" and diagonal blocks" if fused else " below the diagonal blocks") + ", fp64 MFMA 16x16x4)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,  # a committed rocprofv3 --pmc measurement of this command (not this run)
                "flops_per_launch": solve_flops_step * args.steps / max(launches, 1),
                "avg_launch_ms": ms / max(launches, 1),
                "launches": launches,
                "mfma_probe_tflops": probe,  # informative: frac above is against the guide's 78.6
                "mfma_probe_sustained_tflops": probe_sustained,  # the same loop for 1.2 s
                "mfma_probe_sustained_telemetry": probe_tel,
            },
            "stage_ms_per_step": {k: v[0] / args.steps for k, v in fams.items()},
            "step_ms": {"all": step_ms, **(spread(step_ms) or {})},  # HIP events between the steps, on the steps' stream
            "telemetry": telemetry,  # clocks / power / temperatures sampled while the timed steps ran
        }
        if cfg.kernel == "Cholesky":
            facts = 1 + (1 if int((info_keep != 0).sum()) else 0)
            out["roofline_chol"] = roofline_chol(n_arr, facts, fams["chol_gemm"][0] / args.steps, fams["chol_diag"][0] / args.steps, fams["chol_gemm"][1] // max(args.steps, 1))
        tr_a, tr_src = pmc_kernel_traffic("build_A_kernel", args.batch, cfg.name)
        out["roofline_build_A"] = roofline_build_A(n_arr, fams["build_A"][0] / args.steps, tr_a, tr_src,
                                                   samples_on_table=on_table_samples(batch, cfg) if batch is not None else None)
        if farm_out is not None:
            out["farm"] = farm_out
        from pyimcom_amd.blockrun import release_buffers

        def tidy():
            import gc

            torch.cuda.synchronize()
            release_buffers()
            ctx.release_workspace()
            gc.collect()  # (objects in reference cycles hold their tensors until the collector runs)
            torch.cuda.empty_cache()

        def leg(fn):
            """An additional leg must never cost the headline its JSON line: a leg that fails reports the error instead of its numbers."""
            import traceback

            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                traceback.print_exc(file=sys.stderr)
                try:
                    tidy()
                except Exception:  # noqa: BLE001
                    pass
                return {"error": f"{type(e).__name__}: {e}"[:400]}

        if not args.no_block and world == 1 and args.config == "cfg2":
            seams = leg(lambda: seam_legs(ctx, dev, cfg, batch))
            out.update(seams if "error" not in seams else {"seams": seams})
            del batch
            tidy()
            out["block"] = leg(lambda: block_leg(ctx, dev, reps=args.block_reps))
        if not args.no_configs and world == 1 and args.config == "cfg2":
            batch = None
            tidy()
            out["configs"] = leg(lambda: config_legs(ctx, dev))
            out["configs"]["cfg2"] = {"config": "BASELINE configs[1]", "value": out["value"], "unit": out["unit"], "ms_per_stamp": out["ms_per_stamp"],
                                      "roofline": {k: out["roofline"][k] for k in ("kernel", "achieved", "peak", "frac")}, "see": "the top level of this line"}
            # BASELINE configs[2] "batched across one block": the Eigen kernel with its kappa sweep through coadd_block, PSF group per 2 x 2 InStamps
            tidy()  # (the 256-stamp Eigen leg left 130 GB of workspace on the context: the block planner sizes passes by free memory)
            out["eigen_block"] = leg(lambda: block_leg(ctx, dev, n1P=16, reps=1, config="cfg3", warm=8))
            tidy()
            # the reference's own benchmark shape (not a BASELINE config: the shape its users run)
            out["configs"]["paper4"] = leg(lambda: paper4_leg(ctx, dev, cpu_budget=0.0 if args.no_cpu_baseline else min(args.cpu_budget, 25.0)))
            tidy()
            # the reference's DEFAULT configuration: the Iterative kernel at kappa = 0 (71 of its 152 configs)
            out["configs"]["iter_default"] = leg(lambda: iter_default_leg(ctx, dev, cpu_budget=0.0 if args.no_cpu_baseline else min(args.cpu_budget, 20.0)))
        if world == 1 and args.farm:  # (N = 1: the same mosaic on one GPU, for the scaling curve's first point; opt-in)
            batch = None
            tidy()
            out["farm"] = leg(lambda: farm_leg(0, 1, dev, None, "cpu", mosaic=args.farm_mosaic, n1P=args.farm_n1P))
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (rank 0), on a bounded sample
            out["cpu_baseline"] = leg(lambda: cpu_baseline(cfg, cpu_sample, psfs, target, args.cpu_budget))
        out["summary"] = summary_of(out)
        # verbose objects: the detail file and stderr; stdout's LAST line: the compact object a driver parses (< 4 KB)
        sys.stdout.flush()
        print(compact_line(out, write_detail(out)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    raise SystemExit(main())
