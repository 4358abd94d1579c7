"""Block farming on the device (BASELINE configs[3]): `python -m pyimcom_amd.farm` as two processes sharing
cuda:0 (the one-GPU rehearsal of one process per GPU) against a single process, bit for bit; restart behaviour."""

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _farm(out, rank, world, extra=()):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "pyimcom_amd.farm", "--out", out, "--config", "smallm", "--mosaic", "2", "--n1P", "2", "--batch", "3",
           "--shared-gpu", "--schedule", "static", *extra]
    return subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_farm_two_processes_equal_one(tmp_path):
    from pyimcom_amd import farm

    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    procs = [_farm(two, r, 2) for r in range(2)]  # concurrently, on the same GPU
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    p1 = _farm(one, 0, 1)
    o1 = p1.communicate(timeout=600)[0]
    assert p1.returncode == 0, o1
    blocks, costs, _ = farm.synthetic_mosaic("smallm", 2, 2)
    parts = farm.partition(costs, 2)
    assert sorted(parts[0] + parts[1]) == blocks and all(f"done: {sorted(parts[r])}" in outs[r] for r in range(2)), outs
    for b in blocks:
        a, c = np.load(farm.block_path(two, b)), np.load(farm.block_path(one, b))
        assert sorted(a.files) == sorted(c.files)
        for k in a.files:
            assert np.array_equal(a[k], c[k]), (b, k)  # same kernels, same batches: identical bits
        assert np.isfinite(a["out_map"]).all() and np.abs(a["out_map"]).max() > 0 and a["UC"].shape[-1] == 2 * 12
    # restart: every block is skipped; a deleted block alone is redone and comes out identical
    keep = np.load(farm.block_path(one, 2))["out_map"].copy()
    p = _farm(one, 0, 1)
    o = p.communicate(timeout=600)[0]
    assert p.returncode == 0 and o.count("skipped") == 4 and "done: []" in o, o
    os.remove(farm.block_path(one, 2))
    p = _farm(one, 0, 1)
    o = p.communicate(timeout=600)[0]
    assert p.returncode == 0 and o.count("skipped") == 3 and "done: [2]" in o, o
    assert np.array_equal(np.load(farm.block_path(one, 2))["out_map"], keep)


def test_farm_blocks_of_repaired_stamps_with_and_without_hints(tmp_path):
    """The reference's production shape through the farm driver (four blocks of 2 x 2 paper4 stamps in passes of two: every stamp takes
    _cholesky_wrapper's repair, lakernel.py:262-279): what a block's FIRST pass learns about the repair serves the block's other passes
    (blockrun.RepairRecord: expectation and the smallest eigenvalues' whereabouts); every block's first pass starts blind, so a block's
    result does not depend on the blocks before it.  Against the same run with IMCOM_LMIN_HINT=0, in which every smallest-eigenvalue
    iteration starts blind: the same maps to the rounding of the float32 T."""
    from pyimcom_amd import farm

    def run(out, env_extra):
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PYTHONPATH=ROOT, **env_extra)
        cmd = [sys.executable, "-m", "pyimcom_amd.farm", "--out", out, "--config", "paper4", "--mosaic", "2", "--n1P", "2", "--batch", "2", "--schedule", "static"]
        p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:]
        return p.stdout

    hinted, blind = str(tmp_path / "hinted"), str(tmp_path / "blind")
    o1 = run(hinted, {"IMCOM_LMIN_DEBUG": "1"})
    run(blind, {"IMCOM_LMIN_HINT": "0"})
    # the iterations of the hinted run: a pass that starts blind -- every block's first -- needs two factorisations inside its iteration (three
    # when its two stamps reach the closer shift in different rounds), the second pass of every block -- neighbours of the first pass's
    # stamps, started from its record -- ONE
    facs = [int(l.split(" stamps: ")[1].split()[0]) for l in o1.splitlines() if l.startswith("[lmin]") and " stamps: " in l]
    assert len(facs) == 8 and all(2 <= f <= 3 for f in facs[0::2]) and all(f == 1 for f in facs[1::2]), facs
    for b in range(4):
        a, c = np.load(farm.block_path(hinted, b)), np.load(farm.block_path(blind, b))
        assert np.isfinite(a["out_map"]).all() and np.abs(a["out_map"]).max() > 0
        for k in ("out_map", "UC", "Sigma", "kappa"):
            assert np.allclose(a[k], c[k], rtol=2e-5, atol=5e-6 * np.abs(c[k]).max()), (b, k)


def test_farm_shared_block_of_repaired_stamps_is_bit_identical(tmp_path):
    """a block whose stamps all take the Cholesky repair (the reference's production shape), shared between two
    processes WITH hints on, is the single process's block bit for bit: every pass but the block's first starts from the first pass's
    record (a file beside the block's claims), whoever runs it."""
    from pyimcom_amd import farm

    def go(out, rank, world, token):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", PYTHONPATH=ROOT, IMCOM_LMIN_DEBUG="1")
        env.pop("IMCOM_LMIN_HINT", None)
        cmd = [sys.executable, "-m", "pyimcom_amd.farm", "--out", out, "--config", "paper4", "--mosaic", "1", "--n1P", "4", "--batch", "4",
               "--shared-gpu", "--token", token]
        return subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    procs = [go(two, r, 2, "t2") for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    p1 = go(one, 0, 1, "t1")
    o1 = p1.communicate(timeout=900)[0]
    assert p1.returncode == 0, o1
    a, c = np.load(farm.block_path(two, 0)), np.load(farm.block_path(one, 0))
    for k in c.files:
        if not k.startswith("meta_"):
            assert np.array_equal(a[k], c[k]), k
    assert np.isfinite(c["out_map"]).all() and np.abs(c["out_map"]).max() > 0
    # hints were on: in the single process's run the first pass took two factorisations inside its iteration, the three others one
    facs = [int(l.split(" stamps: ")[1].split()[0]) for l in o1.splitlines() if l.startswith("[lmin]") and " stamps: " in l]
    assert 2 <= facs[0] <= 3 and facs[1:] == [1, 1, 1], facs
    assert os.path.exists(os.path.join(two, ".farm-t2", "b0000.repair.json"))


def test_farm_dynamic_schedule_shares_a_block(tmp_path):
    """The dynamic schedule with more ranks than blocks left: ONE block of 6 x 6 stamps in passes of 5 stamps, two processes on
    cuda:0.  The first rank to claim the block plans it; the other joins and takes passes from the end of the plan; the rank that
    finds every pass in part files adds the parts (parity layers: exact), forms the overlap sums in the reference's stamp order,
    recovers the boundary and writes the block -- the single process's block bit for bit (the configuration has fade > 0)."""
    from pyimcom_amd import farm

    def go(out, rank, world, token):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", PYTHONPATH=ROOT)
        cmd = [sys.executable, "-m", "pyimcom_amd.farm", "--out", out, "--config", "small", "--mosaic", "1", "--n1P", "6", "--batch", "5",
               "--shared-gpu", "--token", token, "--psf-groups"]
        return subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    procs = [go(two, r, 2, "t2") for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    p1 = go(one, 0, 1, "t1")
    o1 = p1.communicate(timeout=600)[0]
    assert p1.returncode == 0, o1
    a, c = np.load(farm.block_path(two, 0)), np.load(farm.block_path(one, 0))
    assert sorted(k for k in a.files if k != "meta_ranks") == sorted(k for k in c.files if k != "meta_ranks")
    for k in c.files:
        if k.startswith("meta_"):
            continue
        assert np.array_equal(a[k], c[k]), k  # whoever ran which pass: the single process's bits
    assert np.isfinite(c["out_map"]).all() and np.abs(c["out_map"]).max() > 0
    assert sum(o.count("done: [0]") for o in outs) == 1  # exactly one rank wrote the block
    # both ranks worked on it unless one was so late that every pass was gone
    ranks = a["meta_ranks"].tolist()
    assert ranks in ([0], [1], [0, 1])
    # restart with a new token: the finished block is skipped
    p = go(two, 0, 1, "t3")
    o = p.communicate(timeout=600)[0]
    assert p.returncode == 0 and "skipped" in o and "done: []" in o, o


def test_farm_cfg4_block_with_psf_groups_vs_oracle(tmp_path):
    """ONE real block of the cfg-4 mosaic (BASELINE configs[3]: 48 x 48-output stamps, this block's own exposure depth in
    6 - 10, N = 2.2 - 3.7k) through the farm driver with a PSF group per 2 x 2 InStamps, each group with PSFs of its own
    (`--config cfg4 --psf-groups`), and
    one of its stamps against the oracle: A and -B/2 assembled the reference's way from the groups' PSFOvl objects
    (stamp_system_groups), CholKernel, tapers, coaddition -- compared inside the block file the driver wrote."""
    from oracle import oracle as orc
    from pyimcom_amd import farm, synth
    from pyimcom_amd.blockrun import stamp_neighbours

    out = str(tmp_path / "cfg4")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "pyimcom_amd.farm", "--out", out, "--config", "cfg4", "--mosaic", "1", "--n1P", "2", "--psf-groups", "--seed", "9"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "done: [0]" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    blk = np.load(farm.block_path(out, 0))
    E = int(blk["meta_n_expo"])
    assert 6 <= E <= 10
    # the block's inputs, as the driver built them
    blocks, costs, make_block = farm.synthetic_mosaic("cfg4", 1, 2, 9, True)
    cfg, inst, psfs, target = make_block.host(0)
    n1P, nst, n2 = 2, 4, cfg.n2
    geo = orc.Geom(cfg.npixpsf, cfg.oversamp, cfg.dtheta_as / 3600.0, cfg.flat_penalty)
    # every 2 x 2 group of InStamps has PSFs of its own (synth.group_psfs): the stamp's A draws on the self tables of up to four
    # groups and on the flipped / swapped cross tables between them -- at N = 2.2 - 3.7k against the oracle's PSFOvl objects
    rft_in = {(gj, gi): orc.pad_and_rfft2(synth.group_psfs(psfs, gj, gi), geo) for gj in range(2) for gi in range(2)}
    assert not np.array_equal(rft_in[(0, 0)], rft_in[(1, 0)])
    rft_out = orc.pad_and_rfft2(target, geo)
    C = float(orc.overlap_out_C(rft_out, geo)[0])
    j, i = 2, 1
    ids, pvx, pvy = stamp_neighbours(j, i, n2, nst)
    piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx, pvy)]
    nine = [inst[k] if k >= 0 else None for k in ids]
    sels = [None if t is None else orc.select_pixels(t[0], t[1], pv, cfg.rho) for t, pv in zip(nine, piv)]
    groups = [None if k < 0 else (int(k) // nst >> 1, int(k) % nst >> 1) for k in ids]
    x, y, indata, expo, cum = orc.process_input_stamps(nine, piv, cfg.rho)
    g1 = np.arange(cfg.n2f, dtype=np.float64)
    A, mB = orc.stamp_system_groups(nine, sels, groups, rft_in, rft_out, geo, (i - 1) * n2 + g1, (j - 1) * n2 + g1)
    assert 2000 < A.shape[0] < 4000
    T, UC, Sg, kp, _ = orc.chol_kernel(A, mB, C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax)
    outimage, Tst, Tin, Neff = orc.perform_coaddition(T[None].copy(), indata, expo, E, cfg.n2f, n2, 0, cum)
    ys, xs = slice((j - 1) * n2, j * n2), slice((i - 1) * n2, i * n2)
    lam = np.linalg.eigvalsh(A)
    cond = (lam[-1] + cfg.kappaC[0] * C) / (max(lam[0], 0.0) + cfg.kappaC[0] * C)
    scale = np.abs(T) @ np.abs(indata.T).astype(np.float64)
    err = np.abs(blk["out_map"][0][:, ys, xs].reshape(cfg.n_inframe, -1) - outimage[0].reshape(cfg.n_inframe, -1)) / np.maximum(scale.T, 1e-30)
    assert err.max() <= 2e-5 + 50 * cond * 2.2e-16, err.max()
    s2 = (cfg.n2f, cfg.n2f)
    rt = 1e-5 + 50 * cond * 2.2e-16
    assert np.allclose(blk["UC"][0, ys, xs], UC.reshape(s2), rtol=rt, atol=1e-9) and np.allclose(blk["Sigma"][0, ys, xs], Sg.reshape(s2), rtol=rt, atol=1e-9)
    assert np.allclose(blk["kappa"][0, ys, xs], kp.reshape(s2), rtol=1e-6) and np.allclose(blk["T_weightmap"][0, :, j - 1, i - 1], Tst[0], rtol=2e-5, atol=2e-5 * np.abs(Tst).max())


def test_bench_multi_rank_rehearsal(tmp_path):
    """bench.py's N > 1 path (torch.distributed.run, one rank per GPU, barrier + max-over-ranks timing, rank 0 prints
    the one JSON line) rehearsed on this one-GPU box: two ranks share cuda:0 and rendezvous over gloo.  Not a
    measurement -- it keeps the launch contract of the scaling run from rotting."""
    import json

    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "8",
           "--rehearse-shared-gpu", "--farm-mosaic", "2", "--farm-n1P", "4"]
    detail = str(tmp_path / "detail.json")
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, IMCOM_BENCH_DETAIL=detail), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.rstrip().endswith(lines[0]) and len(lines[0]) < 4096, out.stdout  # ONE compact line, last on stdout
    assert not [l for l in out.stderr.splitlines() if l.startswith("{")]  # (stderr carries the detail as prefixed lines, never a bare JSON line)
    d = json.loads(lines[0])
    full = json.load(open(detail))  # everything verbose: the detail file
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "postage-stamps/s"
    assert d["config"]["stamps_per_step_per_gpu"] == 8 and d["value"] > 0 and "cpu_baseline" not in d and "block" not in d
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # whole-job aggregate over both ranks
    # BASELINE configs[3] on the multi-rank clock: a 2 x 2 mosaic of small cfg-4 blocks farmed over the two ranks (dynamic schedule)
    assert d["farm"]["blocks"] == 4 and d["farm"]["ranks_seen"] == 2 and d["farm"]["value"] > 0 and len(d["per_rank_value"]) == 2
    f = full["farm"]
    assert "error" not in f and f["blocks"] == 4 and f["ranks_seen"] == 2 and f["n_gpus"] == 2 and f["stamps"] == 4 * 16
    assert sum(f["per_rank_blocks_written"]) == 4 and f["value"] > 0 and abs(f["value"] - f["stamps"] / f["makespan_s"]) < 1e-9 * f["value"]
    assert len(f["per_rank_busy_s"]) == 2 and all(0 < b_ <= f["makespan_s"] + 1e-6 for b_ in f["per_rank_busy_s"]) and f["blocks_missing"] == []
    assert d["summary"]["farm"]["v"] > 0 and d["summary"]["farm"]["blocks"] == 4
    # the leg times GPUs: the blocks' host inputs are synthesised before the clock
    # (this rehearsal's mosaic is tiny -- a 0.3 s makespan of which the claim files and polls are a visible part; at the driver's sizes the share
    # is in profiles/r06_*: >= 0.9)
    assert f["host_build_s"] > 0 and f["busy_share"] >= 0.6, (f["busy_share"], f["per_rank_busy_s"], f["makespan_s"])
