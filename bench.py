#!/usr/bin/env python3
"""bench.py — Mpixels/s of the QMF encoder hot path on MI355X.

One "step" = one pass of the hot path over one batch already resident in HBM:
  --config kodak (default; BASELINE.json configs[1]): 256 synthetic 512x768x3 uint8 images -> patch matrices -> exact Gram
      matrices -> SVD initialisation -> 10 BCD iterations -> int8 factors (U, V) in HBM; ranks (7,3,3) = `rank=7`.
  --config clic  (configs[3], per-GPU share): 512 x 1365x2048x3 images per GPU (odd height: 3-row pooling windows, reflect padding).
  --config svd   (configs[4]): lrf.svd_encode's default branch (RGB, X [M,192], uint8-quantised factors) on the kodak batch.
The zlib/JSON container is host work and is not in the timed region.  With N > 1 every rank encodes its own batch (weak
scaling, no data-path collective; `--scaling strong` shards ONE global batch of --batch images over the ranks instead);
RCCL carries only the barrier, the max-reduction of the time and the final stats gather.
Next to `value` (inputs and outputs in HBM) the line carries SURVEY 8(d)'s host->host figure (`host_to_host_mpix_s`,
`pcie_frac`), the decoder (`decode_mpix_s`), the oracle timed on this box's host cores (`cpu_baseline`).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python bench.py --gpus N ...          # no launcher: bench.py starts its own N ranks (fresh child processes, before
                                        # this process has imported torch or touched a GPU) and exits with their code
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
Either way rank 0 refuses to print a line unless the process group really has --gpus ranks (`rccl_ranks` in the line).
`value` is the HBM-resident rate (`value_definition`); the timed region of K steps is repeated `timed_regions` times and
`ms_per_step` / `value` are the MEDIAN region's (min / max beside it).
Before torch is imported every rank binds itself to the CPUs of its GPU's NUMA node (lrf_amd/placement.py; `ranks[].numa`
says what happened; LRF_BENCH_NO_BIND=1 turns it off).  LRF_BENCH_FORCE_DIST=1 initialises the process group at world
size 1 too, so that the RCCL calls of the N > 1 path (init with device_id, float64 all_reduce(MAX) on a device tensor,
barrier, all_gather, destroy) can be executed on a one-GPU box (tests/test_a_two_rank_gpu.py).
SURVEY 8(d) also asks for the rate to final BYTES (lrf.qmf_encode returns bytes, lrf/compression/qmf.py:288-292):
`end_to_end_bytes_mpix_s` (page-locked uint8 batch -> zlib-9 byte streams in host memory, through the pipelined encoder
and liblrf_pack.so on `packer_threads` host threads) and `zlib_ms_per_step` (the container packing alone).
"""
import argparse
import json
import os
import sys
import time

# more hardware queues than the runtime's default of four: the host->host pipeline's upload stream and two kernel streams
# must not share a queue with torch's streams (see lrf_amd/__init__.py); read when the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NUM_ITERS, BOUNDS = 10, (-16, 15)
CONFIGS = {
    "kodak": {"H": 512, "W": 768, "ranks": (7, 3, 3), "batch": 256,
              "label": "{B} x 512x768x3 uint8 per GPU, YCbCr 4:2:0, 8x8 patches, ranks (7,3,3), bounds (-16,15), num_iters 10, "
                       "int8 factors out (BASELINE configs[1])"},
    "clic": {"H": 1365, "W": 2048, "ranks": (7, 3, 3), "batch": 512,
             "label": "{B} x 1365x2048x3 uint8 per GPU (CLIC-sized, odd height), YCbCr 4:2:0, 8x8 patches, ranks (7,3,3), "
                      "bounds (-16,15), num_iters 10, int8 factors out (BASELINE configs[3], one GPU's share of 4096)"},
    "svd": {"H": 512, "W": 768, "ranks": None, "batch": 256, "svd_rank": 5,
            "label": "{B} x 512x768x3 uint8 per GPU, svd_encode default branch (RGB, X [6144,192], quality 2.5 -> R = 5, "
                     "uint8-quantised factors) (BASELINE configs[4])"},
}

# The reference's own torch CPU path, as measured in the build container (it cannot travel to the
# GPU box): quoted in the line so that the ">= 100x the reference" target can be read off it.
REFERENCE_TORCH_CPU = {
    "kodak": {"value": 4.36, "unit": "Mpix/s", "threads": 8, "one_thread": 3.86, "ms_per_image": 90.1,
              "where": "build container, 8-vCPU Xeon 2.1 GHz, torch 2.10 CPU, qmf_encode rank=7 on one 512x768 image "
                       "(BASELINE.md section 2) — a quoted figure, not measured by this run"},
    "clic": {"value": 8.03, "unit": "Mpix/s", "threads": 8, "one_thread": 4.75, "ms_per_image": 348.0,
             "where": "build container, qmf_encode quality=7 on one 1365x2048 image (BASELINE.md section 2; ranks (4,2,2), "
                      "the only CLIC-sized figure measured there) — quoted, not measured by this run"},
}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """threads worth starting: the affinity mask, cut by the cgroup CPU quota when there is one"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, int(q / p + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


_JOB_CORES = None


def _usable_cores_job():
    """usable cores of the whole job as seen when the process started (before any NUMA binding of this rank)"""
    return _JOB_CORES if _JOB_CORES is not None else _usable_cores()


def cpu_baseline(images_u8, H, W, ranks, budget_s=12.0, budget_all_s=10.0):
    """The oracle (CPU port of the reference arithmetic) on a bounded sample of the same batch: one thread (`value`),
    then one thread per usable core (`all_cores`; the oracle is C behind ctypes, which releases the GIL, so plain
    threads scale and no process is started next to the GPU)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle
    oracle.build()

    def encode_one(b):
        X = oracle.rgb_to_planes(images_u8[b])
        for c in range(3):
            oracle.qmf_decompose(X[c], ranks[c], NUM_ITERS, BOUNDS)

    n, t_used, t0 = 0, 0.0, time.perf_counter()
    for b in range(images_u8.shape[0]):
        encode_one(b)
        n += 1
        t_used = time.perf_counter() - t0
        if t_used > budget_s and n >= 8:
            break
    per_image = t_used / n
    cores = _usable_cores()
    n_all = min(max(cores, int(budget_all_s / per_image) * cores), 8 * images_u8.shape[0])
    with ThreadPoolExecutor(max_workers=cores) as pool:
        list(pool.map(encode_one, range(min(cores, images_u8.shape[0]))))  # warm: threads started, pages touched
        t0 = time.perf_counter()
        list(pool.map(lambda i: encode_one(i % images_u8.shape[0]), range(n_all)))
        t_all = time.perf_counter() - t0
    return {"value": round(n * H * W / t_used / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": f"first {n} images of the batch ({H}x{W}x3, ranks {list(ranks)}, {NUM_ITERS} iters), "
                      f"oracle/lrf_oracle.c single thread, {t_used:.1f} s",
            "all_cores": {"value": round(n_all * H * W / t_all / 1e6, 3), "unit": "Mpix/s", "cores": cores,
                          "sample": f"{n_all} images of the same batch on {cores} threads (one image per task), {t_all:.1f} s"},
            "ratio_to_reference_per_thread":
                "the oracle is the reference's arithmetic in scalar C, not the reference's Python: in the build container "
                "(8-vCPU Xeon 2.1 GHz, one thread) it encodes a 512x768 image at ranks (7,3,3) in ~60 ms (43 ms before its Gram "
                "matrix became an exact integer sum) where the reference's torch CPU path takes 91-102 ms (68-90 ms on 8 threads), "
                "i.e. it is ~1.6x FASTER per thread than the reference (outside SURVEY 8(d)'s +-20 % gate, on the conservative "
                "side): GPU/reference ratios are ~1.6x the GPU/oracle ones",
            "cpu": _cpu_model(), "host_cores": os.cpu_count()}


def host_to_host(torch, dist, _lib, dev_index, images, host, H, W, ranks, steps, warmup, use_dist, cdev, repeats=5):
    """SURVEY 8(d)'s metric: uint8 batch in page-locked host memory -> int8 factors back in host memory, through the
    pipelined encoder (lrf_pipe: sub-batches on an upload stream and two kernel streams, H2D / kernels / D2H overlapped).
    With N > 1 every rank runs the leg on its own page-locked buffers (allocated by this rank's thread, the one bound
    to its GPU) between two barriers; the whole-job rate is all ranks' pixels over the slowest rank's time."""
    B = images.shape[0]
    dims = _lib.plane_dims(H, W)
    Uh = torch.empty((B, sum(d[4] * r for d, r in zip(dims, ranks))), dtype=torch.int8, pin_memory=True)
    Vh = torch.empty((B, 64 * sum(ranks)), dtype=torch.int8, pin_memory=True)
    slots = int(os.environ.get("LRF_PIPE_SLOTS", "2"))
    sub = int(os.environ.get("LRF_PIPE_SUB", "0"))
    pipe = _lib.Pipe(dev_index, slots=slots, sub_batch=sub)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def region(fn):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / steps

    def encode():
        pipe.encode_rgb_host(host, ranks, NUM_ITERS, BOUNDS[0], BOUNDS[1], out=(Uh, Vh))

    for _ in range(max(warmup, 2)):
        encode()
    samples = [region(encode) for _ in range(repeats)]
    lo_s, dt, hi_s = timed_stats(samples)
    # the link itself: the same pinned buffer copied to the device with nothing else going on
    dst = torch.empty_like(images)
    for _ in range(2):
        dst.copy_(host, non_blocking=True)
    copies = [region(lambda: dst.copy_(host, non_blocking=True)) for _ in range(3)]
    dt_copy = timed_stats(copies)[1]
    nbytes = host.numel()
    pipe.close()
    del dst
    px = float(B * H * W)
    if use_dist:
        t = torch.tensor([px], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        px = float(t.item())
    return {"host_to_host_mpix_s": round(px / dt / 1e6, 1), "host_to_host_ms_per_step": round(dt * 1e3, 3),
            "host_to_host_ms_min_median_max": [round(x * 1e3, 3) for x in (lo_s, dt, hi_s)],
            "host_to_host_regions": repeats,
            "pcie_frac": round(nbytes / dt / 63e9, 4),
            "pcie_note": "per GPU: input bytes (3 B/pixel, page-locked host memory) / wall time of the whole host->host encode "
                         "(median region, slowest rank), over 63 GB/s (PCIe Gen5 x16); the 0.14 B/pixel of factors return on "
                         "the other direction of the link",
            "h2d_copy_alone_gbs": round(nbytes / dt_copy / 1e9, 2),
            "frac_of_h2d_copy_alone": round(dt_copy / dt, 4),
            "pipe": {"slots": slots, "sub_batch": sub or "auto",
                     "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}}, (Uh, Vh)


def bytes_out_leg(torch, dist, lrf_amd, host, Uh, Vh, H, W, ranks, threads, use_dist, cdev, repeats=3):
    """SURVEY 8(d) "and also end-to-end": the page-locked uint8 batch -> the reference's byte streams (one per image) in
    host memory.  The factorisation streams through the pipelined encoder; the container of each finished piece (metadata
    JSON, per-column zlib level 9, length-prefixed blobs: lrf/compression/utils.py:246-455) is packed by liblrf_pack.so on
    `threads` host threads while the GPU works on the next piece.  Also the packing alone, on factors already in host
    memory.  With N > 1 every rank runs the leg on its own share of the host's cores; the whole-job rate is all ranks'
    pixels over the slowest rank's time."""
    from lrf_amd.codec import pack_streams_native
    B = host.shape[0]

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed(fn):
        sync_all()
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    def encode():
        return lrf_amd.qmf_encode_batch(host, rank=ranks[0], bounds=BOUNDS, num_iters=NUM_ITERS, pack_workers=threads)

    encode()  # warm: packer threads started, pipe buffers allocated
    e2e = [timed(encode) for _ in range(repeats)]
    streams = e2e[-1][1]
    dt = timed_stats([d for d, _ in e2e])[1]
    Un, Vn = Uh.numpy(), Vh.numpy()
    packs = [timed(lambda: pack_streams_native(Un, Vn, (H, W), ranks, BOUNDS, (8, 8), "uint8", threads=threads)) for _ in range(repeats)]
    assert packs[-1][1] == streams, "the pipelined encoder's streams differ from the streams packed from the one-shot factors"
    dt_pack = timed_stats([d for d, _ in packs])[1]
    px = float(B * H * W)
    nbytes = float(sum(len(x) for x in streams))
    if use_dist:
        t = torch.tensor([px, nbytes], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        px, nbytes = float(t[0].item()), float(t[1].item())
    return {"end_to_end_bytes_mpix_s": round(px / dt / 1e6, 1), "end_to_end_bytes_ms_per_step": round(dt * 1e3, 3),
            "zlib_ms_per_step": round(dt_pack * 1e3, 3), "packer_threads": threads,
            "zlib_core_seconds_per_step": round(dt_pack * threads, 4),
            "bits_per_pixel": round(nbytes * 8 / px, 4),
            "end_to_end_note": "page-locked uint8 batch -> one zlib-9 byte stream per image in host memory (what lrf.qmf_encode "
                               "returns), median of %d calls, slowest rank; zlib_ms_per_step = liblrf_pack.so alone on the "
                               "finished factors of the batch (per-column zlib level 9 is host work and the user-visible "
                               "bottleneck: compare host_to_host_ms_per_step; it scales with the host cores the job may use — "
                               "packer_threads is this rank's share of the cgroup quota / affinity mask)" % repeats}


def decode_leg(torch, _lib, ctx, U, V, H, W, ranks, steps):
    """qmf_decode's device part (lrf/compression/qmf.py:329-351) on the factors the encoder just produced."""
    B = U.shape[0]
    for _ in range(3):
        out = ctx.decode_rgb(U, V, H, W, ranks)
    torch.cuda.synchronize()
    ctx.profile_kernels([_lib.LRF_K_DECODE])
    ctx.profile_reset()
    for _ in range(steps):
        out = ctx.decode_rgb(U, V, H, W, ranks)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_time(_lib.LRF_K_DECODE)
    ctx.profile(False)
    alg = 3 * B * H * W + U.numel() + V.numel()  # u8 pixels written once + int8 factors read once
    k = ms / n
    del out
    dims = _lib.plane_dims(H, W)
    left, cleft = (dims[0][3] - dims[0][1]) // 2, (dims[1][3] - dims[1][1]) // 2
    if H % 16 == 0 and W % 16 == 0:
        kname = "k_decode16"
    elif W % 2 == 0 and left % 2 == 0 and (cleft - left // 2) % 4 == 0:  # the host's choice in lrf_qmf_decode_rgb_u8
        kname = "k_decode_strip"
    else:
        kname = "k_decode8"
    return {"decode_mpix_s": round(B * H * W / (k * 1e-3) / 1e6, 1), "decode_ms_per_batch": round(k, 5),
            "decode_roofline": {"bound": "hbm", "kernel": kname,
                                "achieved": round(alg / (k * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                "frac": round(alg / (k * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes_per_launch": alg}}


def smooth_batch(torch, dev, count, H, W, seed=4321):
    """SURVEY 8(d) config 2's smooth variant: bilinear-upsampled (H/8 x W/8) uniform noise + N(0, 4) per pixel, clamped to
    uint8 — low-rank structure, so the clamp / saturation paths, the tie fallback of the Gauss-Seidel and the exact-division
    path see other data than on i.i.d. noise."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    base = torch.rand((count, 3, H // 8, W // 8), device=dev, generator=g) * 255.0
    img = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    img = img + torch.randn((count, 3, H, W), device=dev, generator=g) * 2.0
    return img.clamp_(0, 255).to(torch.uint8)


def sweep_leg(torch, lrf_amd, _lib, ctx, images, H, W, steps=10):
    """The other workloads the reference's sweeps run (experiments/comparison/eval.py:83: luma ranks up to 26; BASELINE config 3:
    24 images per call) next to the headline, each timed like it (inputs and factors in HBM, one region of `steps` steps between
    synchronisations, events on the BCD launches only): quality 7 = ranks (4,2,2) and the smooth-image variant of config 2
    (SURVEY 8(d)), the rank families 9..16 and 17..32 at 256 images, the headline ranks at 24 and 64 images."""
    dims = _lib.plane_dims(H, W)
    B = images.shape[0]
    smooth = smooth_batch(torch, images.device, B, H, W)
    cases = [("256 x 512x768 random, quality 7 = ranks (4,2,2)", images, (4, 2, 2)),
             ("256 x 512x768 smooth (upsampled noise + N(0,4)), ranks (7,3,3)", smooth, (7, 3, 3)),
             ("256 x 512x768 random, ranks (16,8,8)", images, (16, 8, 8)),
             ("256 x 512x768 random, ranks (26,13,13)", images, (26, 13, 13)),
             ("64 x 512x768 random, ranks (7,3,3)", images[:64], (7, 3, 3)),
             ("24 x 512x768 random, ranks (7,3,3) (one quality of BASELINE config 3 in one call)", images[:24], (7, 3, 3))]
    out = []
    for label, imgs, ranks in cases:
        imgs = imgs.contiguous()
        n = imgs.shape[0]
        U = torch.empty((n, sum(d[4] * r for d, r in zip(dims, ranks))), dtype=torch.int8, device=imgs.device)
        V = torch.empty((n, 64 * sum(ranks)), dtype=torch.int8, device=imgs.device)
        for _ in range(3):
            lrf_amd.qmf_factorize_batch(imgs, ranks, NUM_ITERS, BOUNDS, out=(U, V))
        ctx.profile_kernels([_lib.LRF_K_BCD, _lib.LRF_K_BCD_PERSIST])
        ctx.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lrf_amd.qmf_factorize_batch(imgs, ranks, NUM_ITERS, BOUNDS, out=(U, V))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ctx.synchronize()  # (raises if a persistent launch gave up)
        bcd_ms, bcd_n = ctx.kernel_time(_lib.LRF_K_BCD)
        p_ms, p_n = ctx.kernel_time(_lib.LRF_K_BCD_PERSIST)
        ctx.profile(False)
        alg = n * (sum(d[4] for d in dims) * 64 * 4 + sum(d[4] * r for d, r in zip(dims, ranks)))  # one U-update pass over all planes
        if p_n:  # iterations 2..K in one launch — or all K of them (ranks <= 16: no U-update launch of its own is left)
            kname, k_ms, k_bytes = "k_bcd_p", p_ms / p_n, (NUM_ITERS - (1 if bcd_n else 0)) * alg
        else:  # one U-update launch per iteration (a call of several rank families: their launches of an iteration together)
            kname, k_ms, k_bytes = "k_bcd (launch per iteration)", bcd_ms / bcd_n, alg
        out.append({"workload": label, "ranks": list(ranks), "images": n, "ms_per_step": round(dt * 1e3, 4),
                    "mpix_s": round(n * H * W / dt / 1e6, 1),
                    "dominant_kernel": kname, "kernel_avg_ms": round(k_ms, 5), "algorithmic_bytes_per_launch": k_bytes,
                    "roofline_frac": round(k_bytes / (k_ms * 1e-3) / 8e12, 4),
                    "whole_encode_frac_of_hbm_peak": round(75.1 * n * H * W / dt / 8e12, 4)})
        del U, V
    return {"sweep": out,
            "sweep_note": "each entry: HBM-resident like `value`, 3 warm-up calls then one region of %d steps; roofline_frac = "
                          "algorithmic bytes of the dominant BCD launch (X read once per pass + int8 U written once per pass) / its "
                          "HIP-event time / 8 TB/s; whole_encode_frac by SURVEY 8(d)'s 75.1 B/pixel (priced at ranks (7,3,3))" % steps}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n_ranks, argv):
    """`bench.py --gpus N` without a launcher's environment: start the N ranks ourselves, one fresh child process per GPU
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what `python -m torch.distributed.run` would give them.
    Called before this process has imported torch or made any GPU call (a GPU-initialised process must neither fork
    workers nor exec); the parent only waits, forwards rank 0's line (the children inherit stdout) and returns the
    first non-zero exit code, ending the other ranks (by their exact pids) when one fails."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LRF_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # a rank died: the others would wait in a collective for ever
                    q.terminate()
    return rc


def timed_stats(samples_ms):
    s = sorted(samples_ms)
    n = len(s)
    med = s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])
    return s[0], med, s[-1]


def synthetic_batch(torch, dev, lo, count, H, W, base_seed=1234):
    """Images lo .. lo+count-1 of the synthetic dataset, drawn straight into this rank's own batch buffer: image i is
    `randint(0, 256, (3, H, W))` from a generator seeded with base_seed + i, so a rank's block does not depend on the
    number of ranks and no rank ever materialises more than its block (+ nothing: randint writes in place)."""
    images = torch.empty((count, 3, H, W), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    for i in range(count):
        g.manual_seed(base_seed + lo + i)
        torch.randint(0, 256, (3, H, W), dtype=torch.uint8, device=dev, generator=g, out=images[i])
    return images


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="kodak")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = the config's: 256 / 512); with "
                                                         "--scaling strong, the GLOBAL batch sharded over the ranks")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--regions", type=int, default=5, help="how many times the timed region of --steps steps is repeated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host->host, decode and CPU legs")
    args = ap.parse_args()
    assert args.gpus >= 1 and args.steps >= 1 and args.regions >= 1
    global _JOB_CORES
    _JOB_CORES = _usable_cores()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves, BEFORE torch is imported or a GPU touched in this process
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    cfg = CONFIGS[args.config]
    H, W, RANKS = cfg["H"], cfg["W"], cfg["ranks"]
    if os.environ.get("LRF_BENCH_SELF_LAUNCHED") == "1":
        print(f"[bench] rank {os.environ.get('RANK')} of {os.environ.get('WORLD_SIZE')} started (pid {os.getpid()})", file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearsal_env = os.environ.get("LRF_BENCH_REHEARSAL") == "1"
    # CPU placement BEFORE torch is imported: the threads torch / HIP / liblrf_pack.so start and the page-locked buffers this
    # thread touches first then sit on the NUMA node of this rank's GPU (lrf_amd/placement.py, loaded by path: importing
    # the package would import torch)
    if os.environ.get("LRF_BENCH_NO_BIND") == "1" or rehearsal_env:
        binding = {"bound": False, "reason": "off (LRF_BENCH_NO_BIND / rehearsal)", "numa_node": -1, "cpus": _usable_cores()}
        placement = None
    else:
        import importlib.util
        spec = importlib.util.spec_from_file_location("lrf_placement", os.path.join(ROOT, "lrf_amd", "placement.py"))
        placement = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(placement)
        binding = placement.bind_to_gpu_numa(local_rank)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # host threads of this rank's container packer: its share of the cores the job may use (cgroup quota / affinity)
    packer_threads = max(1, min(_usable_cores(), _usable_cores_job() // max(local_world, 1)))

    import torch
    import torch.distributed as dist
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: the launcher's rank count and --gpus must agree"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # LRF_BENCH_REHEARSAL=1: development aid for boxes with fewer GPUs than ranks — ranks share the visible GPUs and the
    # (tiny) collectives run over gloo, because RCCL refuses two ranks on one device.  Never set by the driver.
    rehearsal = os.environ.get("LRF_BENCH_REHEARSAL") == "1"
    n_dev = torch.cuda.device_count()
    assert rehearsal or local_rank < n_dev, f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible"
    dev_index = local_rank % n_dev if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if placement is not None and binding.get("bound"):
        # the sysfs lookup above is an inference (KFD node order = HIP device order): compare with the PCI address the runtime
        # reports for this rank's device and undo a binding made for another GPU's NUMA node (placement.confirm_binding)
        props = torch.cuda.get_device_properties(dev_index)
        bus, slot, dom = (getattr(props, a, None) for a in ("pci_bus_id", "pci_device_id", "pci_domain_id"))
        pci = "%04x:%02x:%02x.0" % (dom or 0, bus, slot or 0) if isinstance(bus, int) else (bus if isinstance(bus, str) else None)
        binding = placement.confirm_binding(binding, pci)
    cdev = torch.device("cpu") if rehearsal else dev  # where the collective payloads live
    backend = None
    force_dist = os.environ.get("LRF_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # forced: a process group of one rank, so that the RCCL code path of the N > 1 runs executes here
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus, f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}"

    import lrf_amd
    from lrf_amd import _lib
    from lrf_amd.sharding import shard_range

    B_cfg = args.batch or cfg["batch"]
    if args.scaling == "strong":  # one global batch, contiguous per-rank blocks (lrf_amd/sharding.py)
        lo, hi = shard_range(B_cfg, rank, world)
        B = hi - lo
    else:  # weak: every rank its own B_cfg images; the dataset is the concatenation of the ranks' blocks
        lo, B = rank * B_cfg, B_cfg
    assert B >= 1, "a rank got no images: --batch must be at least the number of GPUs"
    torch.cuda.reset_peak_memory_stats(dev)
    images = synthetic_batch(torch, dev, lo, B, H, W)
    gen_peak = int(torch.cuda.max_memory_allocated(dev))
    ctx = _lib.context(dev_index)
    dims = _lib.plane_dims(H, W)
    if args.config == "svd":
        R = cfg["svd_rank"]

        def step():
            return ctx.svd_encode_rgb(images, R)
        U = V = None
    else:
        U = torch.empty((B, sum(d[4] * r for d, r in zip(dims, RANKS))), dtype=torch.int8, device=dev)
        V = torch.empty((B, 64 * sum(RANKS)), dtype=torch.int8, device=dev)

        def step():
            lrf_amd.qmf_factorize_batch(images, RANKS, NUM_ITERS, BOUNDS, out=(U, V))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # Set-up, before the W warm-up steps: the first call allocates the workspace and uploads the descriptor tables
    # (16 ms), and the next two still run 10 % slow (first touch of the workspace, clock ramp): three priming calls,
    # disclosed as config.priming_steps.
    PRIMING = 3
    for _ in range(PRIMING):
        step()
    for _ in range(args.warmup):
        step()
    # Inside the timed regions only the dominant kernel (the BCD pass, K launches per step) carries HIP event pairs on
    # the launching stream: an event pair costs stream time (0.15 ms per step when all 22 launches are bracketed).
    ctx.profile_kernels([_lib.LRF_K_BCD, _lib.LRF_K_BCD_PERSIST])
    ctx.profile_reset()
    region_s, own_s = [], []
    for _ in range(args.regions):  # each region: barrier + synchronize, EXACTLY --steps steps, synchronize + barrier
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        own_s.append(time.perf_counter() - t0)  # this rank's own steps, before it waits for the others
        barrier()
        dt_r = time.perf_counter() - t0
        if use_dist:  # the slowest rank's time is the job's time
            t = torch.tensor([dt_r], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        region_s.append(dt_r)
    ctx.profile(False)
    bcd_total_ms, bcd_launches = ctx.kernel_time(_lib.LRF_K_BCD)
    bcdp_total_ms, bcdp_launches = ctx.kernel_time(_lib.LRF_K_BCD_PERSIST)
    # per-kernel breakdown: a separate, untimed pass with every launch bracketed
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ctx.profile(False)
    dt_min, dt, dt_max = timed_stats(region_s)
    rank_stats = None
    own_ms = timed_stats(own_s)[1] / args.steps * 1e3
    my_stats = [float(B * H * W * args.steps), float(dev_index), own_ms, float(binding.get("numa_node", -1)),
                float(binding.get("cpus", 0)), 1.0 if binding.get("bound") else 0.0, float(packer_threads)]
    if use_dist:
        stats = torch.tensor(my_stats, dtype=torch.float64, device=cdev)
        gathered = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(gathered, stats)  # final metrics gather (the only payload collective)
        gathered = [[float(v) for v in s.tolist()] for s in gathered]
    else:
        gathered = [my_stats]
    total_px = sum(s[0] for s in gathered)
    rank_stats = [{"rank": r, "device": int(s[1]), "pixels_per_region": s[0], "ms_per_step": round(s[2], 4),
                   "numa": {"node": int(s[3]), "cpus": int(s[4]), "bound": bool(s[5])}, "packer_threads": int(s[6])}
                  for r, s in enumerate(gathered)]
    rank_stats[rank]["numa"]["reason"] = binding.get("reason", "")

    extras = {}
    h2h_pair = None
    if not args.no_extras and args.config != "svd":  # every rank takes part in the (collective-timed) host->host legs
        Ud, Vd = U.cpu(), V.cpu()
        host = images.cpu().pin_memory()
        h2h, h2h_pair = host_to_host(torch, dist, _lib, dev_index, images, host, H, W, RANKS, min(args.steps, 10), args.warmup,
                                     use_dist, cdev)
        assert torch.equal(h2h_pair[0], Ud) and torch.equal(h2h_pair[1], Vd), "pipelined factors differ from the one-shot ones"
        extras.update(h2h)
        extras.update(bytes_out_leg(torch, dist, lrf_amd, host, Ud, Vd, H, W, RANKS, packer_threads, use_dist, cdev))
        del host

    if rank == 0:
        value = total_px / dt / 1e6
        kern = {}
        for kid, name in _lib.KERNEL_NAMES.items():
            ms, n = ctx.kernel_time(kid)
            if n:
                kern[name] = {"launches_per_step": n / 2, "avg_ms": ms / n, "ms_per_step": ms / 2}
        ms_step = dt / args.steps * 1e3
        if args.config == "svd":
            metric = "Mpixels/sec svd_encode (RGB, 8x8 patch, quality 2.5, uint8 factors)"
            M = dims[0][4]
            # whole step: u8 in (3 B/px) + X [M,192] written once and read twice (Gram, U = X w) as BYTES since round 3 (3 B/px
            # each; fp32 until round 2: 12 B/px each) + factors
            alg_bytes = B * (3 * H * W + 3 * M * 192 + (M + 192) * cfg["svd_rank"])
            achieved = alg_bytes / (ms_step * 1e-3) / 1e9
            roof = {"bound": "hbm",
                    "note": "the step is paced by the per-matrix eigen-solver chain (256 matrices of 192 x 192, one workgroup "
                            "each: k_any_tridiag_reg + k_any_eig are two thirds of it), not by bytes; the HBM figure is for completeness",
                    "kernel": "svd_encode (all kernels of the step)", "achieved": round(achieved, 1),
                    "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(ms_step, 5)}
        else:
            metric = "Mpixels/sec qmf_encode (8x8 patch, r=7, 10 iters)"
            # dominant kernel: one BCD pass (k_bcd_w).  Algorithmic bytes per launch (DESIGN.md "Roofline"):
            # X read once (4 B per patch element) + int8 U written once.
            alg_bytes = B * (sum(d[4] for d in dims) * 64 * 4 + sum(d[4] * r for d, r in zip(dims, RANKS)))
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if bcdp_launches:
                # the iterations run in ONE launch (k_bcd_p: the U-update passes pulled from a queue, the V updates inside):
                # all K of them since round 5 (no k_bcd launch is seen then), iterations 2..K before; algorithmic bytes of that
                # launch = that many passes
                first_inside = bcd_launches == 0
                passes = NUM_ITERS if first_inside else NUM_ITERS - 1
                bcd_ms = bcdp_total_ms / bcdp_launches  # live, from the timed regions (rank 0's GPU)
                achieved = passes * alg_bytes / (bcd_ms * 1e-3) / 1e9
                if args.config == "kodak" and B == 256 and os.path.exists(tfile):
                    try:
                        traffic = json.load(open(tfile)).get("k_bcd_p", {}).get("hbm_bytes_per_launch")
                    except (ValueError, OSError):
                        traffic = None
                roof = {"bound": "hbm", "kernel": "k_bcd_p", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": passes * alg_bytes, "avg_launch_ms": round(bcd_ms, 5),
                        "launches_measured": bcdp_launches,
                        "traffic_source": ("profiles/traffic_latest.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command on the same sources (tools/run_profiles_r05.sh, tools/make_traffic.py) — a committed "
                                           "measurement, not collected in this run") if traffic is not None else None,
                        "note": ("k_bcd_p = iterations 1..%d in one launch (%d U-update passes of %d B each + the per-matrix V updates, "
                                 "which the launch-per-iteration path ran as k_vupdate)" % (NUM_ITERS, passes, alg_bytes)) if first_inside else
                                ("k_bcd_p = iterations 2..%d in one launch (%d U-update passes of %d B each + the per-matrix V updates, "
                                 "which the launch-per-iteration path ran as k_vupdate); the first iteration is k_bcd_w<1>: "
                                 "%.5f ms per launch" % (NUM_ITERS, passes, alg_bytes, bcd_total_ms / max(bcd_launches, 1)))}
            else:
                bcd_ms = bcd_total_ms / bcd_launches  # live, from the timed regions (rank 0's GPU)
                achieved = alg_bytes / (bcd_ms * 1e-3) / 1e9
                if args.config == "kodak" and B == 256 and os.path.exists(tfile):
                    try:
                        traffic = json.load(open(tfile)).get("k_bcd", {}).get("hbm_bytes_per_launch")
                    except (ValueError, OSError):
                        traffic = None
                roof = {"bound": "hbm", "kernel": "k_bcd_w", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(bcd_ms, 5),
                        "launches_measured": bcd_launches,
                        "traffic_source": ("profiles/traffic_latest.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command on the same sources (tools/run_profiles_r05.sh, tools/make_traffic.py) — a committed "
                                           "measurement, not collected in this run") if traffic is not None else None}
        out = {
            "metric": metric,
            "value": round(value, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["label"].format(B=B), "priming_steps": PRIMING,
                       "global_batch": B_cfg if args.scaling == "strong" else B * world,
                       "images_this_rank": [lo, lo + B], "rank_input_bytes": images.numel(),
                       "input_generation_peak_bytes": gen_peak,
                       "parallelism": f"images sharded over {world} GPU(s), no data-path collective"},
            "value_definition": "hbm_resident: uint8 batch and int8 factors stay in HBM across the timed region "
                                "(SURVEY 8(d)'s pinned-host -> host rate is host_to_host_mpix_s)",
            "timed_regions": args.regions,
            "ms_per_step_min": round(dt_min / args.steps * 1e3, 4),
            "ms_per_step_median": round(ms_step, 4),
            "ms_per_step_max": round(dt_max / args.steps * 1e3, 4),
            "timing_note": "each region = barrier + synchronize, exactly `steps` steps, synchronize + barrier, max over ranks; "
                           "value / ms_per_step are the median region's",
            "rccl_ranks": world, "collective_backend": backend, "self_launched": os.environ.get("LRF_BENCH_SELF_LAUNCHED") == "1",
            "roofline": roof,
            # 75.1 algorithmic bytes per input pixel for the whole encode at K = 10
            "whole_encode_frac_of_hbm_peak": round(75.1 * total_px / world / dt / 8e12, 4) if args.config != "svd" else None,
            "kernels": {k: {a: round(b, 5) for a, b in v.items()} for k, v in kern.items()},
            "kernels_note": "per-kernel breakdown from a separate untimed pass with every launch bracketed by events (k_bcd_persist = "
                            "the iterations in one launch where k_bcd_p runs — all K of them at ranks <= 16, the V updates inside; "
                            "k_bcd / k_vupdate appear only where an iteration is a launch of its own; k_planes_gram = k_planes16_gram, the patch "
                            "matrices and the luma planes' Gram partials in one kernel, k_gram then covers the chroma planes only); roofline.avg_launch_ms is "
                            "measured inside the timed regions (events on the BCD launches only)",
        }
        out["ranks"] = rank_stats
        out["forced_dist"] = force_dist
        out.update(extras)
        if world == 1 and not args.no_extras and args.config != "svd":
            out.update(decode_leg(torch, _lib, ctx, U, V, H, W, RANKS, args.steps))
            if args.config == "kodak" and B == 256:
                out.update(sweep_leg(torch, lrf_amd, _lib, ctx, images, H, W))
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(images[:min(B, 256)].cpu().numpy(), H, W, RANKS)
                out["cpu_baseline"]["reference_torch_cpu"] = REFERENCE_TORCH_CPU.get(args.config)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
