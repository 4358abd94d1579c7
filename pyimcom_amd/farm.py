"""Block farming over the GPUs of a node: one process per GPU, blocks claimed on start (or a static partition), no data-path
collective.

The reference's only parallelism is one OS process per block (docs/run_README.rst:81-100,
examples/multiblock_norep.pl:42-66); blocks share read-only inputs and write separate outputs, so ranks never
exchange data.  torch.distributed is used by callers only for start/end barriers.
"""

from typing import List, Sequence


def estimate_cost(n_pixels: float, m: int, nv: int = 1) -> float:
    """Relative cost of one stamp: factor N^3/3 + solve 2 N^2 m per kappa node."""
    return nv * (n_pixels**3 / 3.0 + 2.0 * n_pixels**2 * m)


def partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of units (blocks) to `world` ranks.

    Deterministic (ties by index), so every rank computes the same partition without communicating."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def my_units(costs: Sequence[float], rank: int, world: int) -> List[int]:
    return partition(costs, world)[rank]


def batches(units: Sequence[int], batch: int) -> List[List[int]]:
    """Split a rank's units into device batches (the last one may be short)."""
    return [list(units[i : i + batch]) for i in range(0, len(units), batch)]


# ---------------------------------------------------------------------------------------------------------------
# The driver: blocks -> ranks -> coadd_block -> one output file per block.
#
# The reference coadds a mosaic by starting one OS process per block, skipping blocks whose output file is already
# there (examples/multiblock_norep.pl:25-27, 42-66; docs/run_README.rst:81-100).  Here one process per GPU walks its
# share of the block list; a block's inputs come from a `make_block(b)` callable, so the same driver serves the
# synthetic mosaics of the benchmarks and a maintainer's real blocks (make_block = what Block.__init__ prepares,
# see pyimcom_amd.refblock).  Ranks never exchange data.
def block_path(outdir, b):
    import os

    return os.path.join(outdir, f"block_{int(b):04d}.npz")


def write_block(path, maps, meta=None):
    """One .npz per block, written under a temporary name and renamed, so that a killed run never leaves a file that
    a restart would mistake for a finished block."""
    import os

    import numpy as np

    out = {"out_map": maps.out_map.cpu().numpy(), "T_weightmap": maps.T_weightmap.cpu().numpy()}
    out.update({k: v.cpu().numpy() for k, v in maps.maps.items()})
    for k, v in (meta or {}).items():
        out["meta_" + k] = np.asarray(v)
    _save_npz(path, out)


# ---- dynamic schedule: blocks claimed on start, the passes of a block shared when whole blocks run out ------------------
#
# Coordination is by files in a hidden directory of the output directory (atomic O_EXCL creation = a claim; atomic rename =
# a result) -- the reference's own mechanism, generalised: its block runner skips a block whose output file exists
# (examples/multiblock_norep.pl:25-27).  Nothing is exchanged between ranks in memory, and no rank waits for another while
# there is work it could do.
#
#   .farm-<token>/b<id>.claim            a rank has taken block <id> (it builds the inputs, plans the passes, writes the plan)
#   .farm-<token>/b<id>.plan.json        the block's passes (lists of stamps): helpers run the same plan
#   .farm-<token>/b<id>.c<q>.claim       pass q of the block is taken
#   .farm-<token>/b<id>.part.<rank>.<k>.npz  what a rank has coadded of a SHARED block (BlockMaps.state(): arrays that add
#                                        exactly between ranks) + the list of its passes; a block one rank coadds alone has none
#   .farm-<token>/b<id>.merge            the rank that found every pass of the block in part files and sums them
#   .farm-<token>/b<id>.done             the block's file has been written by this launch
#
# A rank walks the blocks in order of decreasing cost and claims the first free one (list scheduling: what a static LPT
# partition computes in advance, but from the real durations); when no unclaimed block is left it joins the blocks still in
# progress, most remaining work first, and takes passes from the END of their plans while their owners walk from the front.
# The parts of a shared block add up EXACTLY (with fade > 0 they hold the stamps' tiles apart in parity layers, block.py), the
# sums of overlapping stamps are formed once, in the reference's stamp order, by the rank that merges: a shared block is the
# single process's block bit for bit, whoever ran which pass.
# Every claim names its owner (rank, pid, the process's start time, host).  A claim whose owner no longer exists on this host
# -- a killed launch that used the same <token>, a rank that died in this one -- is taken over (atomic rename, one winner), so
# stale claims never hide a block; a rank that has nothing left to do stays until every block file exists, taking over what
# dying ranks leave behind, and run() raises if blocks remain that nobody works on.
# <token> identifies the launch: IMCOM_FARM_RUN, else a value broadcast by rank 0 when torch.distributed is initialised, else
# the process itself (one rank) or its parent (pid and start time; the ranks of one torch.distributed.run share it).
def _try_create(path, text=""):
    import os

    try:
        fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
    except FileExistsError:
        return False
    with os.fdopen(fd, "w") as f:
        f.write(text)
    return True


def _atomic_write(path, write):
    import os

    tmp = f"{path}.tmp{os.getpid()}"
    write(tmp)
    os.replace(tmp, path)


def _save_npz(path, arrays):
    """np.savez under a temporary name (which no reader's pattern matches), then renamed: a reader never sees a partial file."""
    import os

    import numpy as np

    d, f = os.path.split(path)
    tmp = os.path.join(d, f".tmp{os.getpid()}.{f}")
    with open(tmp, "wb") as fh:
        np.savez(fh, **arrays)
    os.replace(tmp, path)


def _proc_start(pid):
    """Start time of process `pid` in clock ticks since boot (/proc/<pid>/stat field 22); None if there is no such process
    (or no /proc): with the pid it names ONE process, whatever the kernel does with the number later."""
    try:
        with open(f"/proc/{int(pid)}/stat") as f:
            s = f.read()
        fields = s[s.rindex(")") + 2 :].split()
        return None if fields[0] in "ZX" else fields[19]  # a zombie (exited, not yet reaped by its parent) works no more
    except (OSError, ValueError, IndexError):
        return None


def _ids():
    """What tells two processes on machines of the same name apart: the boot id of the kernel and the PID namespace (containers with host
    networking share a hostname but not their process tables)."""
    import os

    def rd(fn, f):
        try:
            return fn(f).strip() if fn is not os.readlink else fn(f)
        except OSError:
            return None

    return rd(lambda f: open(f).read(), "/proc/sys/kernel/random/boot_id"), rd(os.readlink, "/proc/self/ns/pid")


def _owner(rank):
    import json
    import os
    import socket

    boot, pidns = _ids()
    return json.dumps({"rank": int(rank), "pid": os.getpid(), "start": _proc_start(os.getpid()), "host": socket.gethostname(), "boot": boot, "pidns": pidns})


STALE_AFTER = 1800.0  # seconds without a heartbeat after which a claim whose owner cannot be checked from here counts as abandoned (IMCOM_FARM_STALE_S)


def _gen_path(path, g):
    """Generation g of a claim: `path` itself, then path.t1, path.t2, ... -- a claim is never renamed or removed; taking over from a
    dead owner is the (atomic, exclusive) creation of the NEXT generation, so two takers cannot both win and nobody's fresh claim can
    be swept away by a taker that judged its predecessor (the rename-based takeover had a check-then-rename race)."""
    return path if g == 0 else f"{path}.t{g}"


def _latest_gen(path):
    import os

    if not os.path.exists(path):
        return -1
    g = 0
    while os.path.exists(_gen_path(path, g + 1)):
        g += 1
    return g


def _one_claim_state(path, grace=10.0):
    """State of ONE generation file.  None: no such file; True: its owner is (or may be) alive; False: the owner is gone."""
    import json
    import os
    import socket
    import time

    try:
        with open(path) as f:
            text = f.read()
        rec = json.loads(text) if text.strip() else None
    except FileNotFoundError:
        return None
    except (OSError, ValueError):
        rec = None
    if not isinstance(rec, dict) or "pid" not in rec:
        # being written right now (created, not yet filled), or written by a process that died in between
        try:
            return time.time() - os.path.getmtime(path) < grace
        except OSError:
            return None
    boot, pidns = _ids()
    same_table = (rec.get("host") == socket.gethostname() and rec.get("start") is not None and _proc_start(os.getpid()) is not None
                  and rec.get("boot", boot) == boot and rec.get("pidns", pidns) == pidns)
    if not same_table:
        # another machine, another boot, another PID namespace: the owner's process cannot be looked up from here.  It is taken to be
        # alive while it keeps touching its claim (owners do, at every pass: _heartbeat) and abandoned after STALE_AFTER seconds of silence
        try:
            return time.time() - os.path.getmtime(path) < float(os.environ.get("IMCOM_FARM_STALE_S", STALE_AFTER))
        except OSError:
            return None
    return _proc_start(rec["pid"]) == rec["start"]


def _claim_state(path, grace=10.0):
    """None: no such claim; True: the owner of its latest generation is (or may be) alive; False: that owner is gone -- the claim is stale."""
    g = _latest_gen(path)
    return None if g < 0 else _one_claim_state(_gen_path(path, g), grace)


def _claim(path, me):
    """Claim `path` for the owner record `me`: create it, or -- when the owner of its latest generation no longer exists -- create
    the next generation.  Exactly one of any number of concurrent callers gets True."""
    import os

    g = 0
    while True:
        if _try_create(_gen_path(path, g), me):
            return True
        if os.path.exists(_gen_path(path, g + 1)):
            g += 1  # somebody has taken this generation over already: judge the next one
            continue
        if _one_claim_state(_gen_path(path, g)) is False:
            g += 1  # a dead owner: try to be the one who creates the next generation
            continue
        return False


def _heartbeat(path, me=None):
    """Touch the latest generation of a claim (for readers that cannot look its owner's process up) -- only when that generation's
    record is this process's own (``me``: its owner record): a helper that shares a block's tail passes must not keep a dead
    cross-host owner's block claim looking alive."""
    import json
    import os

    g = _latest_gen(path)
    if g < 0:
        return False
    p = _gen_path(path, g)
    if me is not None:
        try:
            with open(p) as f:
                rec, mine = json.loads(f.read() or "null"), json.loads(me)
            if not isinstance(rec, dict) or any(rec.get(k) != mine.get(k) for k in ("rank", "pid", "start", "host")):
                return False
        except (OSError, ValueError):
            return False
    try:
        os.utime(p)
        return True
    except OSError:
        return False


def _is_claim_file(name):
    import re

    return re.search(r"\.(claim|merge)(\.t\d+)?$", name) is not None


def launch_token(world, token=None):
    """The name of this launch's coordination directory (see above)."""
    import os

    if token is not None:
        return str(token)
    if os.environ.get("IMCOM_FARM_RUN"):
        return os.environ["IMCOM_FARM_RUN"]
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() == world and world > 1:
            import uuid

            box = [uuid.uuid4().hex[:12] if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return f"d{box[0]}"
    except ImportError:
        pass
    pid = os.getpid() if world == 1 else os.getppid()
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    return f"{pid}.{_proc_start(pid) or 0}" + (f".{run_id}" if world > 1 and run_id != "none" else "")


class _GpuBackend:
    """How the driver touches the device; the host-logic tests substitute plain-Python stand-ins."""

    def __init__(self, batch):
        self.batch = batch

    def plan(self, spec):
        from .blockrun import plan_block

        return plan_block(spec["cfg"], spec["pool"], spec["tables"], spec["n1P"], self.batch)

    def coadd(self, spec, chunks, claim):
        """-> ({name: array}, passes run): what this process coadded of the block, as arrays that add exactly between
        processes (BlockMaps.state()); nothing is summed over overlapping stamps or recovered at the boundary yet."""
        import torch

        from .blockrun import coadd_block

        # what the block's FIRST pass saw of the Cholesky repair serves its other passes, whoever runs them (blockrun.RepairRecord; the
        # driver keeps it in a file beside the block's claims): a shared block has the single process's bits
        maps = coadd_block(spec["cfg"], spec["pool"], spec["tables"], spec["n1P"], spec["n_expo"], chunks=chunks, claim=claim, pad_sides=None,
                           repair_record=spec.get("_repair_record"), first_chunk=spec.get("_first_chunk", 0))
        torch.cuda.synchronize()
        return maps.state(), list(maps.chunks_done)

    def finalize(self, spec, arrays):
        """The summed states -> the block's arrays: overlap sums in the reference's stamp order, boundary recovery
        (coadd.py:2163-2181)."""
        import torch

        from .block import BlockMaps

        cfg = spec["cfg"]
        n_out = arrays["T_weightmap"].shape[0]
        maps = BlockMaps(spec["n1P"], cfg.n2, cfg.fade, cfg.n_inframe, spec["n_expo"], ctx=spec["tables"].ctx, device=str(spec["pool"].device), n_out=n_out)
        maps.load_state(arrays)
        maps.finalize(spec.get("pad_sides", ""), spec.get("postage_pad", 0))
        torch.cuda.synchronize()
        return maps.arrays()


class _FileRepairRecord:
    """blockrun.RepairRecord in a file of the launch's claim directory (b<id>.repair.json): written once, atomically, by the process that
    ran the block's first pass; the other processes' passes wait for it (they join a block from the end of its plan, normally long after
    its first pass is through).  ``max_wait`` seconds without the file: the first pass's owner is taken to be gone -- an error, as a
    pass would otherwise start from something else than what the single process's pass starts from."""

    def __init__(self, path, poll=0.05, max_wait=None):
        self.path, self.poll, self.max_wait, self.value = path, poll, max_wait, None

    def put(self, share, hint):
        import json

        self.value = {"share": float(share), "hint": None if hint is None else float(hint)}
        text = json.dumps(self.value)

        def wr(tmp):
            with open(tmp, "w") as f:
                f.write(text)

        _atomic_write(self.path, wr)

    def get(self):
        import json
        import time

        if self.value is not None:
            return self.value
        t0 = time.monotonic()
        while True:
            try:
                with open(self.path) as f:
                    self.value = json.load(f)
                return self.value
            except (OSError, ValueError):
                pass
            if self.max_wait is not None and time.monotonic() - t0 > self.max_wait:
                raise RuntimeError(f"{self.path}: the block's first pass left no repair record within {self.max_wait:.0f} s")
            time.sleep(self.poll)


class _Prefetch:
    """make_block(b) split in a host part and a device part (optional attributes ``make_block.host(b)`` and
    ``make_block.device(b, host_data)``): the host part of the block a rank will probably take next runs on a thread while
    the current block is on the GPU."""

    def __init__(self, make_block, enabled):
        self.mk = make_block
        self.split = enabled and hasattr(make_block, "host") and hasattr(make_block, "device")
        self.b, self.thread, self.box = None, None, {}

    def start(self, b):
        import threading

        if not self.split or b is None or self.b == b:
            return
        self.join()
        self.b, self.box = b, {}

        def work(b=b, box=self.box):
            try:
                box["host"] = self.mk.host(b)
            except Exception as exc:  # re-raised by get() on the main thread
                box["error"] = exc

        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def join(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None

    def get(self, b):
        if not self.split:
            return self.mk(b)
        if self.b == b:
            self.join()
            box, self.b = self.box, None
            if "error" in box:
                raise box["error"]
            return self.mk.device(b, box["host"])
        return self.mk.device(b, self.mk.host(b))


def run(blocks, costs, make_block, outdir, rank=0, world=1, batch=None, device=None, restart=True, log=print, coadd=None,
        schedule="dynamic", token=None, backend=None, prefetch=True, poll=0.05, max_wait=None, stats=None):
    """Coadd this rank's share of `blocks` (ids) and write block_<id>.npz files into `outdir`.

    costs[k]: relative cost of blocks[k] (estimate_cost summed over its stamps);
    make_block(b) -> dict(cfg=, pool=, tables=, n1P=, n_expo=, [pad_sides=, postage_pad=, meta=]) -- the arguments of
    pyimcom_amd.blockrun.coadd_block -- called on a rank that works on b, right before it starts;
    restart: skip blocks whose file exists (multiblock_norep.pl:25-27).
    schedule "dynamic" (default): blocks are claimed on start in order of decreasing cost and the passes of the last blocks
    are shared (see above); "static": the longest-processing-time partition of `costs`, computed identically by every rank
    -- no files besides the outputs.  `coadd` (static) / `backend` (dynamic) replace the device calls (the host-logic tests run
    the driver without a GPU that way).  ``max_wait`` (seconds, optional): a rank with nothing left to take raises when the blocks it
    waits for -- in the hands of ranks it cannot look up -- make no progress for that long, instead of waiting for ever (a claim whose
    owner cannot be checked counts as abandoned after IMCOM_FARM_STALE_S = 1800 s without a heartbeat and is then taken over).
    ``stats`` (optional dict): filled with this rank's busy_s / wall_s / blocks_written / passes_run.
    Returns the list of block ids whose output file this rank wrote."""
    import json
    import os
    import time
    import zipfile

    import numpy as np

    os.makedirs(outdir, exist_ok=True)
    on_gpu = coadd is None and backend is None
    if on_gpu:
        import torch

        dev = device or f"cuda:{rank % max(torch.cuda.device_count(), 1)}"
        torch.cuda.set_device(dev)
    t_start, busy = time.perf_counter(), 0.0
    done = []
    if schedule == "static":
        if coadd is None:
            from .blockrun import coadd_block as coadd
        for b in [blocks[k] for k in my_units(costs, rank, world)]:
            path = block_path(outdir, b)
            if restart and os.path.exists(path):
                log(f"[farm rank {rank}] block {b}: {path} exists, skipped")
                continue
            t0 = time.perf_counter()
            spec = make_block(b)
            maps = coadd(spec["cfg"], spec["pool"], spec["tables"], spec["n1P"], spec["n_expo"], batch=batch,
                         pad_sides=spec.get("pad_sides", ""), postage_pad=spec.get("postage_pad", 0))
            if on_gpu:
                torch.cuda.synchronize()
            write_block(path, maps, spec.get("meta"))
            done.append(b)
            busy += time.perf_counter() - t0
            log(f"[farm rank {rank}] block {b}: {spec['n1P'] ** 2} stamps, {spec['n_expo']} exposures, {time.perf_counter() - t0:.2f} s -> {path}")
        log(f"[farm rank {rank}] busy {busy:.2f} s of {time.perf_counter() - t_start:.2f} s (static schedule)")
        if stats is not None:
            stats.update(busy_s=busy, wall_s=time.perf_counter() - t_start, blocks_written=len(done), passes_run=None)
        return done
    assert schedule == "dynamic"
    import re

    be = backend or _GpuBackend(batch)
    token = launch_token(world, token)
    cdir = os.path.join(outdir, f".farm-{token}")
    os.makedirs(cdir, exist_ok=True)
    _sweep_dead_launches(outdir, cdir)
    me = _owner(rank)
    cp = lambda b, what: os.path.join(cdir, f"b{int(b):04d}.{what}")  # noqa: E731
    order = [blocks[k] for k in sorted(range(len(blocks)), key=lambda k: (-costs[k], k))]
    cost_of = dict(zip(blocks, costs))
    pre = _Prefetch(make_block, prefetch)
    nparts, passes_run = [0], [0]

    def finished(b):  # by this launch, or (restart) by an earlier one
        return os.path.exists(block_path(outdir, b)) and (restart or os.path.exists(cp(b, "done")))

    def take(path):
        """Claim `path`: create it, or take it over from an owner that no longer exists."""
        return _claim(path, me)

    def claimable(path):
        return _claim_state(path) in (None, False)

    def free_blocks():
        return [b for b in order if not finished(b) and claimable(cp(b, "claim"))]

    passes_memo = {}  # part file name -> its passes (a part file is complete and immutable once its name exists: written under another name, renamed)

    def part_passes(b):
        """[(name, passes)] of the readable part files of block b, in the order of their first pass.  Only the small `passes` member of a
        file is read, once (np.load is lazy per member): ranks that wait for the last shared block poll this 20 times a second, and a part
        holds the block's maps in four parity layers -- hundreds of MB at production size."""
        pat = re.compile(rf"^b{int(b):04d}\.part\.\d+\.\d+\.npz$")
        out = []
        for f in sorted(os.listdir(cdir)):
            if not pat.match(f):
                continue
            if f not in passes_memo:
                try:
                    with np.load(os.path.join(cdir, f)) as z:
                        passes_memo[f] = [int(q) for q in z["passes"]]
                except (OSError, ValueError, EOFError, KeyError, zipfile.BadZipFile):
                    continue  # not a complete file (yet): its passes do not count
            out.append((f, passes_memo[f]))
        out.sort(key=lambda t: min(t[1]))
        return out

    def parts_of(b):
        """The same with the arrays loaded: [(name, passes, arrays)] (the merge, once the pass list is complete and the merge claim taken)."""
        out = []
        for f, ps in part_passes(b):
            with np.load(os.path.join(cdir, f)) as z:
                out.append((f, ps, {k: z[k] for k in z.files if k != "passes"}))
        return out

    def covered(b):
        return sorted(q for _, ps in part_passes(b) for q in ps)

    def write_out(b, spec, total, ranks):
        out = be.finalize(spec, total)
        for k, v in (spec.get("meta") or {}).items():
            out["meta_" + k] = np.asarray(v)
        out["meta_ranks"] = np.asarray(sorted(ranks))
        _save_npz(block_path(outdir, b), out)
        _try_create(cp(b, "done"), me)

    def run_passes(b, spec, chunks, from_end):
        """Claim and run passes of block b.  The whole block by this rank alone: written right away; else the part file.
        -> (passes run, block file written)"""
        nonlocal busy
        idx = list(range(len(chunks)))
        if from_end:
            idx.reverse()
        view = [chunks[q] for q in idx]
        t0 = time.perf_counter()

        held = [None]  # the pass claim this process took last

        def claim(k):
            path = cp(b, f"c{idx[k]:04d}.claim")
            _heartbeat(cp(b, "claim"), me)  # (for ranks that cannot look this process up: the block's owner -- if that is this process -- is at work)
            if held[0] is not None:
                _heartbeat(held[0], me)  # and the pass claim this process holds
            if _try_create(path, me):
                held[0] = path
                return True
            # the pass of a rank that died before it wrote its part
            return _claim_state(path) is False and idx[k] not in covered(b) and _claim(path, me)

        spec["_repair_record"] = _FileRepairRecord(cp(b, "repair.json"), poll, max_wait if max_wait is not None else 1800.0)
        spec["_first_chunk"] = idx.index(0)  # where the block's first pass sits in this view of its plan
        arrays, ran = be.coadd(spec, view, claim)
        ran = [idx[k] for k in ran]
        passes_run[0] += len(ran)
        wrote = False
        if len(ran) == len(chunks) and take(cp(b, "merge")):
            write_out(b, spec, arrays, [rank])
            wrote = True
        elif ran:
            _save_npz(cp(b, f"part.{rank}.{nparts[0]}.npz"), dict(arrays, passes=np.asarray(ran)))
            nparts[0] += 1
        busy += time.perf_counter() - t0
        return ran, wrote

    def try_merge(b, spec, nchunks):
        nonlocal busy
        have = covered(b)
        if len(set(have)) != len(have):
            raise RuntimeError(f"block {b}: passes {have} were coadded more than once")
        if have != list(range(nchunks)) or finished(b) or not take(cp(b, "merge")):
            return False
        t0 = time.perf_counter()
        parts = parts_of(b)  # the arrays are read here, by the one rank that merges
        total = {k: v.copy() for k, v in parts[0][2].items()}
        for _, _, arrays in parts[1:]:  # in the order of the parts' first passes
            for k in total:
                total[k] += arrays[k]
        write_out(b, spec, total, {int(f.split(".")[2]) for f, _, _ in parts})
        busy += time.perf_counter() - t0
        return True

    def load_plan(b):
        try:
            with open(cp(b, "plan.json")) as f:
                return [[tuple(t) for t in c] for c in json.load(f)]
        except (OSError, ValueError):
            return None

    def phase1():
        """Whole blocks, largest first.  -> something was done"""
        did = False
        while True:
            mine = next((b for b in free_blocks() if take(cp(b, "claim"))), None)
            if mine is None:
                return did
            did = True
            t0 = time.perf_counter()
            spec = pre.get(mine)
            rest = [b for b in free_blocks() if b != mine]
            pre.start(rest[0] if rest else None)  # the host part of the block this rank will most likely take next
            chunks = load_plan(mine)  # left by an owner that died: its helpers and their parts follow that plan
            if chunks is None:
                chunks = be.plan(spec)
                _atomic_write(cp(mine, "plan.json"), lambda tmp: json.dump([[list(map(int, t)) for t in c] for c in chunks], open(tmp, "w")))
            ran, wrote = run_passes(mine, spec, chunks, from_end=False)
            wrote = wrote or try_merge(mine, spec, len(chunks))
            if wrote:
                done.append(mine)
            log(f"[farm rank {rank}] block {mine}: {len(ran)} of {len(chunks)} passes, {spec['n_expo']} exposures, {time.perf_counter() - t0:.2f} s"
                + (f" -> {block_path(outdir, mine)}" if wrote else " (shared: merged by the rank that finishes last)"))

    def phase2():
        """Help with the blocks in progress, most remaining work first; merge what dying ranks left complete.  -> something was done"""
        did, tried = False, set()
        while True:
            open_blocks = []
            for b in order:
                if b in tried or finished(b) or _claim_state(cp(b, "merge")):
                    continue
                chunks = load_plan(b)
                if chunks is None:
                    continue
                have = covered(b)
                left = [q for q in range(len(chunks)) if q not in have and claimable(cp(b, f"c{q:04d}.claim"))]
                if left or have == list(range(len(chunks))):
                    open_blocks.append((cost_of[b] * len(left) / len(chunks), -order.index(b), b, chunks, left))
            if not open_blocks:
                return did
            _, _, b, chunks, left = max(open_blocks, key=lambda t: t[:2])
            tried.add(b)
            t0 = time.perf_counter()
            spec = pre.get(b)
            ran, wrote = run_passes(b, spec, chunks, from_end=True) if left else ([], False)
            wrote = wrote or try_merge(b, spec, len(chunks))
            did = did or bool(ran) or wrote
            if wrote:
                done.append(b)
            log(f"[farm rank {rank}] block {b}: helped with {len(ran)} of {len(chunks)} passes, {time.perf_counter() - t0:.2f} s"
                + (f" -> {block_path(outdir, b)}" if wrote else ""))

    for b in order:
        if restart and os.path.exists(block_path(outdir, b)):
            log(f"[farm rank {rank}] block {b}: {block_path(outdir, b)} exists, skipped")
    waiting, waiting_since = None, time.perf_counter()
    while True:
        did = phase1()
        did = phase2() or did
        pending = [b for b in order if not finished(b)]
        if not pending:
            break
        if did:
            continue
        # nothing this rank could take: either other ranks are at work on the rest (stay: if one of them dies its claims go
        # stale and are taken over above), or the rest is lost
        def live(b):
            names = [f for f in os.listdir(cdir) if f.startswith(f"b{int(b):04d}.") and _is_claim_file(f)]
            return any(_one_claim_state(os.path.join(cdir, f)) for f in names)

        lost = [b for b in pending if not live(b)]
        if lost and lost == [b for b in lost if not finished(b) and not live(b)]:  # (looked twice: a merge may have ended in between)
            raise RuntimeError(f"[farm rank {rank}] blocks {lost} have no output file and nobody works on them")
        if waiting != pending:
            log(f"[farm rank {rank}] nothing left to take; blocks {pending} are in other ranks' hands")
            waiting, waiting_since = pending, time.perf_counter()
        if max_wait is not None and time.perf_counter() - waiting_since > max_wait:
            raise RuntimeError(f"[farm rank {rank}] waited {max_wait:.0f} s for blocks {pending} in other ranks' hands without any of them finishing (max_wait)")
        time.sleep(poll)
    pre.join()
    wall = time.perf_counter() - t_start
    log(f"[farm rank {rank}] busy {busy:.2f} s of {wall:.2f} s ({100.0 * busy / max(wall, 1e-9):.0f} %), wrote {len(done)} block files")
    if stats is not None:
        stats.update(busy_s=busy, wall_s=wall, blocks_written=len(done), passes_run=passes_run[0])
    return done


def _sweep_dead_launches(outdir, keep):
    """Remove the coordination directories of earlier launches in which no claim has a live owner (a killed launch leaves one
    behind; its finished blocks are the block files, which stay)."""
    import os
    import shutil

    import time

    for d in os.listdir(outdir):
        path = os.path.join(outdir, d)
        if not d.startswith(".farm-") or path == keep or not os.path.isdir(path):
            continue
        try:
            if time.time() - os.path.getmtime(path) < 60.0:  # (a launch that has only just made its directory has no claims yet)
                continue
            names = [f for f in os.listdir(path) if _is_claim_file(f)]
            if not any(_one_claim_state(os.path.join(path, f)) for f in names):
                shutil.rmtree(path, ignore_errors=True)
        except OSError:
            pass


def simulate(costs, passes, world, overhead=0.0):
    """Makespan of the dynamic schedule for blocks of cost `costs[k]` split in `passes[k]` equal passes on `world` ranks, every
    rank following run()'s rules (largest free block first; then the block with the most work left, passes from the end; a
    helper pays `overhead` -- building the block's inputs -- when it joins a block).  -> (makespan, per-rank busy time)"""
    import heapq

    n = len(costs)
    order = sorted(range(n), key=lambda k: (-costs[k], k))
    front, back = [0] * n, list(passes)  # [front, back): the passes nobody has taken (owner from the front, helpers from the end)
    owner = {}
    busy = [0.0] * world
    ev = [(0.0, r, None) for r in range(world)]  # (time a rank becomes free, rank, block it is working on)
    heapq.heapify(ev)
    end = 0.0
    while ev:
        t, r, cur = heapq.heappop(ev)
        end = max(end, t)
        dt = None
        if cur is not None and front[cur] < back[cur]:  # stay on the block: next pass
            if owner[cur] == r:
                front[cur] += 1
            else:
                back[cur] -= 1
            dt = costs[cur] / passes[cur]
        else:
            nxt = next((k for k in order if k not in owner), None)
            if nxt is not None:  # a whole block, largest first
                owner[nxt] = r
                front[nxt] += 1
                cur, dt = nxt, costs[nxt] / passes[nxt]
            else:  # help: the block with the most work left
                left = [(costs[k] * (back[k] - front[k]) / passes[k], -order.index(k), k) for k in owner if front[k] < back[k]]
                if left:
                    cur = max(left)[2]
                    back[cur] -= 1
                    dt = overhead + costs[cur] / passes[cur]
        if dt is not None:
            busy[r] += dt
            heapq.heappush(ev, (t + dt, r, cur))
    return end, busy


def synthetic_mosaic(config="cfg4", nblock=4, n1P=2, seed=4, psf_groups=False):
    """The cfg-4 workload of BASELINE.json as a block list: nblock x nblock blocks of n1P x n1P output stamps, every
    block with its own exposure depth (uniform in the configuration's range), lattices and data.  Returns
    (blocks, costs, make_block)."""
    import dataclasses

    import numpy as np

    from . import synth

    base = synth.CONFIGS[config]
    rng = np.random.default_rng(seed)
    lo, hi = base.n_expo if isinstance(base.n_expo, tuple) else (base.n_expo, base.n_expo)
    depth = rng.integers(lo, hi + 1, nblock * nblock)
    blocks = list(range(nblock * nblock))
    p = synth.NATIVE_ARCSEC / base.dtheta_as
    n_pix = lambda e: e * (base.n2 + 2 * base.rho) ** 2 / p**2  # noqa: E731  input pixels of one stamp, roughly
    costs = [n1P * n1P * estimate_cost(n_pix(int(e)), base.m, len(base.kappaC)) for e in depth]

    def host(b):
        """The host part of a block's inputs (InStamps, sampled PSFs): what a prefetch thread can prepare."""
        E = int(depth[b])
        cfg = dataclasses.replace(base, n_expo=E, name=f"{base.name}_b{b}")
        inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng([seed, b]))
        psfs, target = synth.make_psfs(cfg, E, seed=20260723 + b)
        return cfg, inst, psfs, target

    def device_part(b, host_data, device="cuda:0"):
        import torch

        from ._lib import default_context
        from .select import InStampPool
        from .stamps import BlockTables, PSFGroupTables

        cfg, inst, psfs, target = host_data
        ctx = default_context(torch.device(device).index or 0)  # one context per GPU
        E = int(depth[b])
        if psf_groups:
            ng = (n1P + 3) // 2
            tables = BlockTables({(gj, gi): synth.group_psfs(psfs, gj, gi) for gj in range(ng) for gi in range(ng)}, target, cfg.nfft, ctx=ctx,
                                 device=device, cells=True)  # every group its own PSFs: self, cross and input-output tables all differ
        else:
            tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=device)
        return dict(cfg=cfg, pool=InStampPool(inst, cfg.n_inframe, device=device), tables=tables, n1P=n1P, n_expo=E,
                    meta=dict(block=b, n_expo=E, config=config))

    def make_block(b, device="cuda:0"):
        return device_part(b, host(b), device)

    make_block.host, make_block.device = host, device_part
    return blocks, costs, make_block


def main(argv=None):
    """python -m pyimcom_amd.farm --out DIR [--config cfg4 --mosaic 4 --n1P 2]: coadd a synthetic mosaic, this process
    taking the blocks of rank RANK of WORLD_SIZE (environment, as torch.distributed.run sets them; default 0 of 1).
    One process per GPU (LOCAL_RANK); --shared-gpu puts every rank on cuda:0 (a rehearsal on a one-GPU box)."""
    import argparse
    import os

    ap = argparse.ArgumentParser(prog="python -m pyimcom_amd.farm", description=main.__doc__)
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--mosaic", type=int, default=4, help="blocks per side")
    ap.add_argument("--n1P", type=int, default=2, help="output stamps per block side")
    ap.add_argument("--batch", type=int, default=None, help="stamps per pass (default: blockrun.choose_batch)")
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--psf-groups", action="store_true", help="a PSF group per 2x2 InStamps (BlockTables) instead of one per block")
    ap.add_argument("--no-restart", action="store_true", help="recompute blocks whose output exists")
    ap.add_argument("--shared-gpu", action="store_true")
    ap.add_argument("--schedule", choices=("dynamic", "static"), default="dynamic",
                    help="dynamic: blocks claimed on start, passes of the last blocks shared; static: LPT partition of the estimated costs")
    ap.add_argument("--token", default=None, help="identifies the launch for the dynamic schedule's claim files (default: the parent process id)")
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.shared_gpu else int(os.environ.get("LOCAL_RANK", str(rank)))
    blocks, costs, make_block = synthetic_mosaic(a.config, a.mosaic, a.n1P, a.seed, a.psf_groups)
    dev = f"cuda:{local}"
    mk = lambda b: make_block(b, dev)  # noqa: E731
    mk.host, mk.device = make_block.host, (lambda b, h: make_block.device(b, h, dev))
    done = run(blocks, costs, mk, a.out, rank, world, a.batch, device=dev, restart=not a.no_restart, schedule=a.schedule, token=a.token)
    print(f"[farm rank {rank}/{world}] done: {sorted(done)}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
