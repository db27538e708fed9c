#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: images/sec of one MM-UNet training step
(forward + Dice+BCE loss + backward + DP gradient all-reduce + AdamW) on synthetic 3x512x512 batches,
bs=8 per GPU, data-parallel over N GPUs of one node (one process per GPU, RCCL over xGMI).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  "roofline":     the dominant hand-written kernel (selective-scan forward at the headline shape
                  B=8, D=128, L=65536, N=16) timed live with HIP events on the stream it
                  is launched on; achieved = algorithmic bytes s*B*L*(4D+2N) / mean duration of one
                  mmu_selective_scan_fwd call; peak = 8 TB/s HBM3E; traffic from profiles/ if measured;
  "roofline_bwd", "roofline_d6", "roofline_bwd_d6", "roofline_conv1d", "roofline_conv1d_bwd":
                  the same for the scan backward (bytes s*B*L*(8D+2N) + 4*B*L*2N), for both at the 6-channel census
                  shape of MMConv's Mamba blocks (B=8, D=6, L=65536), and for causal conv1d (2 / 3 * s*B*D*L);
  "roofline_noz", "roofline_bwd_noz": the scan at the headline shape with z = NULL -- what the three large blocks call since
                  the gate moved into csrc/tri_fused.hip (bytes s*B*L*(3D+2N) / s*B*L*(5D+2N) + 4*B*L*2N);
                  "roofline_tri_*": that file's four streaming kernels (4 / 5 / 9 / 5 [B, D, L] streams);
  "roofline_gemm_nt": the token-contraction weight-gradient product (csrc/gemm_nt_splitk.hip) at out_proj's shape of the
                  128-channel Mamba blocks (128 x 64 over B*L = 524,288 tokens), bytes s*(M+N)*B*L, same timing method;
  "roofline_conv": the MFMA-bound kernel of the path (csrc/conv3x3_mfma.hip at CBAM's shape [8,64,256,256] 64->64 and
                  at a Unet shape [8,256,64,64] 256->256), same timing method; achieved = bf16 MFMA FLOP/s actually
                  issued (3 passes of 2*B*Cout*H*W*Cin*9 for the hi/lo split); peak = 2.5 PFLOP/s dense bf16;
  "cpu_baseline": the CPU oracle (oracle/model_ref.py + C scan, kind "port") running the same training
                  step (fwd + loss + bwd) on one 3x512x512 image on this host's cores (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE: 8)")
    ap.add_argument("--size", type=int, default=512, help="image side (BASELINE: 512)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="compute dtype; f32 = what the reference's train.py runs (no autocast)")
    ap.add_argument("--d-state", type=int, default=16, help="Mamba state size (BASELINE config 5: 64)")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--infer", action="store_true",
                    help="forward-only (BASELINE config 2) instead of the training step; prints its own metric name")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (the command profiles/r01_roofline_leg_kernel_stats.csv is taken from)")
    ap.add_argument("--cpu-size", type=int, default=512, help="image side of the cpu_baseline sample")
    return ap.parse_args()


def _timed(dev, fn, iters):
    """Mean duration (ms) of fn() over `iters` calls, HIP events on the stream the kernels are launched on."""
    for _ in range(3):
        fn()
    st = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def _traffic(name):
    tpath = os.path.join(ROOT, "profiles", name)
    if os.path.exists(tpath):
        try:
            return json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def _scan_case(dev, b, d, l, n, layout):
    """Inputs with the reference test's distributions (tests/ops/test_selective_scan.py:53-88).  layout "dbl": u, delta,
    z, dout physically [D][B][L] (what mamba_inner hands the kernels); "bdl": contiguous."""
    gen = torch.Generator(device=dev).manual_seed(0)
    A = -0.5 * torch.rand(d, n, device=dev, generator=gen)
    B = torch.randn(b, 1, n, l, device=dev, generator=gen)
    C = torch.randn(b, 1, n, l, device=dev, generator=gen)
    D = torch.randn(d, device=dev, generator=gen)
    bias = 0.5 * torch.rand(d, device=dev, generator=gen)

    def mk(scale_rand=False):
        shape = (d, b, l) if layout == "dbl" else (b, d, l)
        t = 0.5 * torch.rand(*shape, device=dev, generator=gen) if scale_rand else torch.randn(*shape, device=dev, generator=gen)
        return t.permute(1, 0, 2) if layout == "dbl" else t
    u, z, dout = mk(), mk(), mk()
    delta = mk(True)
    return u, delta, A, B, C, D, z, bias, dout


def _scan_issue_floor_ms(b, d, l, n, backward):
    """Issue-time floor of the selective scan at fp32 on gfx950: per element group (64 (channel, state,
    token) elements = one wave-instruction each) the recurrence needs one v_exp_f32 (4.29 ns of a SIMD at two waves per
    SIMD, profiles/r02_valu_rates_microbench.txt) and delta * A, (delta u) B, h = a h + b, y += C h as four halves of
    packed FMAs (2.46 ns per v_pk_fma_f32): 9.2 ns -> 151 us at the headline shape on 1,024 SIMDs.  Backward: the
    forward again, the adjoint recurrence and the gradient streams -- about 2.5 x (a second exp is not needed by the
    algorithm, 14 packed halves are)."""
    groups = b * d * n * l / 64.0
    per_group_ns = 4.29 + (4 if not backward else 14) * 2.46 / 2.0
    return groups * per_group_ns / 1024 / 1e6


def scan_rooflines(dev, iters=20):
    """Live measurement of the hand-written streaming kernels against the HBM roofline (algorithmic bytes of
    SURVEY.md 8d / DESIGN.md 4): selective-scan forward and backward at the headline shape, the same at the
    6-channel census shape (MMConv's Mamba blocks), and causal conv1d forward / backward."""
    from mm_unet_amd import causal_conv1d_hip as cc, selective_scan_hip as ss
    s_ = 4  # fp32
    legs = {}

    def leg(name, kernel, shape, alg_bytes, ms, binding, traffic=None, issue_floor_ms=None):
        ach = alg_bytes / (ms * 1e-3) / 1e9
        legs[name] = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                      "frac": round(ach / 8000.0, 4), "traffic": traffic, "kernel": kernel, "binding": binding,
                      "shape": shape, "algorithmic_bytes": alg_bytes, "ms_per_launch": round(ms, 4)}
        if issue_floor_ms is not None:
            # the second roof: the instruction-issue time of the ALGORITHM's minimum mix at the VALU
            # rates measured on this chip (profiles/r02_valu_rates_microbench.txt), next to the HBM fraction
            legs[name]["issue_floor_ms"] = round(issue_floor_ms, 4)
            legs[name]["frac_of_issue_floor"] = round(issue_floor_ms / ms, 4)

    for (b, d, l, n, tag) in ((8, 128, 65536, 16, ""), (8, 6, 65536, 16, "_d6")):
        # both legs on the layout the model hands the kernels: u / delta / z / dout physically [D][B][L]
        u, delta, A, B, C, D, z, bias, dout = _scan_case(dev, b, d, l, n, "dbl")
        shape = {"batch": b, "dim": d, "seqlen": l, "dstate": n, "dtype": "f32"}
        ms_f = _timed(dev, lambda: ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=False), iters)
        # the backward as mamba_inner calls it: the forward's un-gated `out` is kept and handed over (the reference's
        # backward reads it too, selective_scan.cpp:338) -- 8 D streams: u, delta, z, dout, out in; du, ddelta, dz out
        out, x = ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=True)[:2]
        ms_b = _timed(dev, lambda: ss.bwd(u, delta, A, B, C, D, z, bias, dout, x, out, None, True, False), iters)
        del out
        streaming = b * d >= 512
        leg("roofline" + tag,
            "mmu_selective_scan_fwd = scan_fwd_stream16_kernel (one launch: every (batch, channel) row scanned front to "
            "back, state carried in registers; 16 tokens per lane, two state pairs per wave in its half-waves)" if streaming else
            "mmu_selective_scan_fwd = chunk_reduce8 + chunk_carry_par + chunk_apply_fwd8 (chunk-parallel: too few "
            "rows to stream)", shape, s_ * b * l * (4 * d + 2 * n), ms_f,
            "the vector pipe, not HBM: one v_exp_f32 + ~7 VALU instructions per (d, n, t) element group (466 + 84 "
            "transcendental per wave and 512-token tile); the time follows the instruction count (DESIGN.md 4.1)", _traffic("scan_fwd_traffic.json") if tag == "" else None,
            issue_floor_ms=_scan_issue_floor_ms(b, d, l, n, backward=False))
        leg("roofline_bwd" + tag,
            "mmu_selective_scan_bwd = chunk_reduce8<bwd> + chunk_carry_par + chunk_apply_bwd_w8 (512-token tiles, one "
            "state pair per wave, 4-step row scans on the chunk carries, the forward's `out` read) + reduce_partials_w8",
            shape, s_ * b * l * (8 * d + 2 * n) + 4 * b * l * 2 * n, ms_b,
            "VALU issue: ~15 packed fp32 + 2 exp + 6.6 DPP instructions per (d, state pair, t) in the apply kernel "
            "(recompute + adjoint scan + 8 gradient streams) at ~2.5 ns per packed instruction and SIMD, DESIGN.md 4.2",
            _traffic("scan_bwd_traffic.json") if tag == "" else None,
            issue_floor_ms=_scan_issue_floor_ms(b, d, l, n, backward=True))
        if tag == "":
            # since f2 (tri_inner.py) the three large blocks call the scan WITHOUT z: u, delta in, y out (3 D streams);
            # backward u, delta, dout in, du, ddelta out (5) + the fp32 dB / dC rows
            y_nz, x_nz = ss.fwd(u, delta, A, B, C, D, None, bias, True)[:2]
            ms_fn = _timed(dev, lambda: ss.fwd(u, delta, A, B, C, D, None, bias, True), iters)
            ms_bn = _timed(dev, lambda: ss.bwd(u, delta, A, B, C, D, None, bias, dout, x_nz, None, None, True, False), iters)
            del y_nz
            leg("roofline_noz", "mmu_selective_scan_fwd with z = NULL (scan_fwd_stream16_kernel<HAS_Z = false>): the call "
                "the tri-directional block makes since the gate moved into tri_gate (csrc/tri_fused.hip)", shape,
                s_ * b * l * (3 * d + 2 * n), ms_fn, "the vector pipe (as `roofline`)",
                issue_floor_ms=_scan_issue_floor_ms(b, d, l, n, backward=False))
            leg("roofline_bwd_noz", "mmu_selective_scan_bwd with z = NULL (chunk_reduce8<bwd> + chunk_carry_par + "
                "chunk_apply_bwd_w8<HAS_Z = false> + reduce_partials_w8)", shape,
                s_ * b * l * (5 * d + 2 * n) + 4 * b * l * 2 * n, ms_bn, "VALU issue (as `roofline_bwd`)",
                issue_floor_ms=_scan_issue_floor_ms(b, d, l, n, backward=True))
            # f2's own four kernels at the largest block (64 slices): each a pure stream
            from mm_unet_amd import tri_inner
            xz = torch.randn(2 * d, b, l, device=dev).permute(1, 0, 2)
            cws = [torch.randn(d, 4, device=dev) for _ in range(3)]
            cbs = [torch.randn(d, device=dev) for _ in range(3)]
            ys = tri_inner.tri_conv_fwd(xz[:, :d], 64, cws, cbs)
            dxz = torch.empty_like(xz)
            tshape = {"batch": b, "dim": d, "seqlen": l, "nslices": 64, "dtype": "f32"}
            unit = s_ * b * d * l
            for nm, k, fn in (("tri_conv_fwd", 4, lambda: tri_inner.tri_conv_fwd(xz[:, :d], 64, cws, cbs)),
                              ("tri_gate_fwd", 5, lambda: tri_inner.tri_gate_fwd(xz[:, d:], 64, ys)),
                              ("tri_gate_bwd", 9, lambda: tri_inner.tri_gate_bwd(xz[:, d:], 64, ys, dout, dxz[:, d:])),
                              ("tri_conv_bwd", 5, lambda: tri_inner.tri_conv_bwd(xz[:, :d], 64, cws, cbs, ys, dxz[:, :d]))):
                leg("roofline_" + nm, "mmu_" + nm + " (csrc/tri_fused.hip)", tshape, k * unit, _timed(dev, fn, iters),
                    "HBM (streaming, %d [B, D, L] streams)" % k if nm != "tri_conv_bwd" else
                    "instruction issue: three conv pre-activations, three silu', 12 + 12 FMAs per token at 2 workgroups per CU")
            del xz, ys, dxz
            w = torch.randn(d, 4, device=dev)
            cb = torch.randn(d, device=dev)
            ms_cf = _timed(dev, lambda: cc.causal_conv1d_fwd(u, w, cb, True), iters)
            ms_cb = _timed(dev, lambda: cc.causal_conv1d_bwd(u, w, cb, dout, None, True), iters)
            cshape = {"batch": b, "dim": d, "seqlen": l, "width": 4, "dtype": "f32"}
            leg("roofline_conv1d", "mmu_causal_conv1d_fwd (width 4 + bias + SiLU)", cshape, 2 * s_ * b * d * l, ms_cf,
                "HBM (streaming)")
            leg("roofline_conv1d_bwd", "mmu_causal_conv1d_bwd (dx, dW, db)", cshape, 3 * s_ * b * d * l, ms_cb,
                "HBM (streaming) + per-block reductions for dW / db")
            # the projections' weight gradients: dW = G . X^T contracted over all B * L tokens;
            # out_proj's shape of the 128-channel Mamba blocks, operands as the model holds them ([C][B][L] storage)
            from mm_unet_amd import mfma_gemm
            gm, gn = 128, 64
            ga = torch.randn(gm, b, l, device=dev)
            gb = torch.randn(gn, b, l, device=dev)
            ms_nt = _timed(dev, lambda: mfma_gemm.gemm_nt(ga, gb, gm, gn, b, l, b * l, l, b * l, l), iters)
            leg("roofline_gemm_nt", "mmu_gemm_nt_splitk = gemm_nt_wide_kernel (128-token steps, bf16 hi/lo split on the "
                "matrix cores, fp32 accumulate) + gemm_nt_reduce_kernel (ordered slab sums)",
                {"m": gm, "n": gn, "tokens": b * l, "dtype": "f32"}, s_ * (gm + gn) * b * l, ms_nt,
                "HBM (streaming: both operands read once, 512 contiguous bytes per row and load)",
                _traffic("gemm_nt_traffic.json"))
            del ga, gb
    return legs


def scan_roofline(dev, iters=20):
    return scan_rooflines(dev, iters)["roofline"]


def conv_roofline(dev, iters=20):
    """Live measurement of the dense 3x3 convolution on the matrix cores (forward)."""
    from mm_unet_amd.conv3x3_mfma import conv3x3_mfma
    out = {"bound": "mfma", "peak": 2500.0, "unit": "TFLOP/s", "kernel": "conv3x3_mfma_kernel (bf16 hi/lo split, "
           "3 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate)", "traffic": None, "shapes": []}
    gen = torch.Generator(device=dev).manual_seed(0)
    for (b, cin, cout, h, w) in ((8, 64, 64, 256, 256), (8, 256, 256, 64, 64)):
        x = torch.randn(b, cin, h, w, device=dev, generator=gen)
        wt = torch.randn(cout, cin, 3, 3, device=dev, generator=gen) / (3 * cin ** 0.5)
        for _ in range(3):
            conv3x3_mfma(x, wt, None)
        st = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            conv3x3_mfma(x, wt, None)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters          # includes the weight-preparation kernel of each call
        alg = 2.0 * b * cout * h * w * cin * 9
        out["shapes"].append({"input": [b, cin, h, w], "out_channels": cout, "ms_per_launch": round(ms, 4),
                              "algorithmic_flops": alg, "fp32_grade_tflops": round(alg / (ms * 1e-3) / 1e12, 1),
                              "achieved": round(3 * alg / (ms * 1e-3) / 1e12, 1),
                              "frac": round(3 * alg / (ms * 1e-3) / 1e12 / 2500.0, 4)})
    # headline = the shape MM_Net runs (CBAM, 64 -> 64 at 256 x 256); the second one is the Unet model's.
    # `achieved` counts the three bf16 MFMA passes of the hi/lo split (what the matrix cores issue); by algorithmic
    # FLOPs the kernel delivers `fp32_grade_tflops`, ~2x the fp32-MFMA peak of 157 TFLOP/s
    best = out["shapes"][0]
    out["achieved"], out["frac"] = best["achieved"], best["frac"]
    out["headline_shape"] = "MM_Net CBAM: 8 x 64 x 256 x 256 -> 64 (MMUNet.py:313-338)"
    tpath = os.path.join(ROOT, "profiles", "conv3x3_mfma_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            for s_ in out["shapes"]:
                key = "x".join(str(v) for v in s_["input"]) + "->" + str(s_["out_channels"])
                s_["traffic"] = tj.get(key, {}).get("hbm_bytes_per_launch")
            out["traffic"] = best.get("traffic")
        except Exception:
            pass
    return out


def cpu_baseline(size):
    """The oracle's training step (fwd + Dice+BCE + bwd) on ONE image, timed on this host's cores."""
    import oracle
    from oracle import model_ref
    from mm_unet_amd.mmunet import MM_Net
    oracle.build()
    torch.manual_seed(50)
    sd = {k: v.detach().clone() for k, v in MM_Net(num_classes=1).state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_()
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, size, size, generator=gen)
    t = (torch.rand(1, 1, size, size, generator=gen) > 0.88).float()
    t0 = time.time()
    logits = model_ref.mm_net(sd, x, training=True)
    model_ref.dice_bce_loss(logits, t).backward()
    dt = time.time() - t0
    cores = max(torch.get_num_threads(), oracle.num_threads())
    return {"value": round(1.0 / dt, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle/model_ref.py MM_Net fwd + Dice+BCE + bwd on 1 image 3x{size}x{size} fp32 "
                      f"({dt:.1f} s, torch CPU ops + C scan/conv1d oracle, {cores} threads)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal of the N > 1 path on a one-GPU box (never a measurement): every rank on device 0, gloo instead of RCCL
    # (RCCL refuses two ranks on one device):  MMUNET_BENCH_REHEARSAL=1 python -m torch.distributed.run ... bench.py
    rehearsal = os.environ.get("MMUNET_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.roofline_only:
        legs = scan_rooflines(dev)
        legs["roofline_conv"] = conv_roofline(dev)
        print(json.dumps(legs), flush=True)
        return

    if args.infer:
        from mm_unet_amd.mmunet import MM_Net
        from mm_unet_amd.train_step import InferStep
        torch.manual_seed(50)
        step = InferStep(MM_Net(num_classes=1, d_state=args.d_state).to(dev), amp_dtype=torch.bfloat16 if args.dtype == "bf16" else None,
                         use_graph=not args.no_graph)
        gen = torch.Generator(device=dev).manual_seed(1000 + rank)
        images = torch.randn(args.batch, 3, args.size, args.size, device=dev, generator=gen)
        for _ in range(args.warmup + 2):
            step(images)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(images)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        assert torch.isfinite(out).all()
        print(json.dumps({"metric": "images/sec forward (inference), MM-UNet 3x512x512 bs=8 per GPU",
                          "value": round(args.batch * args.steps / dt, 3), "unit": "images/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                          "higher_is_better": True, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": f"MM_Net eval forward, 3x{args.size}x{args.size}, bs={args.batch}",
                                     "launch": "eager" if args.no_graph else "hip-graph replay"}}), flush=True)
        return
    from mm_unet_amd.dp import broadcast_module_state
    from mm_unet_amd.loss import DICE_BCE_Loss
    from mm_unet_amd.mmunet import MM_Net
    from mm_unet_amd.train_step import TrainStep, make_optimizer

    torch.manual_seed(50)  # reference: same_seeds(50), train.py:160
    model = MM_Net(num_classes=1, d_state=args.d_state).to(dev).train()
    if world > 1:
        broadcast_module_state(model)
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    graph = not args.no_graph
    step = TrainStep(model, DICE_BCE_Loss(), make_optimizer(model, capturable=graph), amp_dtype=amp, use_graph=graph)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)  # rank r owns samples [8r, 8r+8)
    images = torch.randn(args.batch, 3, args.size, args.size, device=dev, generator=gen)
    targets = (torch.rand(args.batch, 1, args.size, args.size, device=dev, generator=gen) > 0.88).float()

    for _ in range(args.warmup + (3 if graph else 0)):   # graph mode: 2 eager warm-up steps + capture
        step(images, targets)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(images, targets)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    assert torch.isfinite(loss), "training diverged in the benchmark"

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        line = {
            "metric": f"images/sec fwd+bwd, MM-UNet 3x{args.size}x{args.size} bs={args.batch} per GPU",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"MM_Net train step (fwd + Dice+BCE + bwd + grad all-reduce + AdamW), "
                                   f"3x{args.size}x{args.size}, bs={args.batch}/GPU, random-init seed 50"
                                   + (f", d_state={args.d_state}" if args.d_state != 16 else ""),
                       "global_batch": args.batch * world, "image": [3, args.size, args.size],
                       "parallelism": f"dp{world}", "grad_allreduce_bytes": step.reducer.payload_bytes(),
                       "launch": "hip-graph replay" if graph else "eager",
                       # what the timed step consists of: on one rank ONE replayed graph holds forward, loss, backward and
                       # AdamW; on several ranks the graph ends with the backward pass, the gradient exchange (pack ->
                       # RCCL all-reduce -> unpack, 3 buckets) and the table-driven AdamW follow it as eager launches
                       # -- so the per-rank step at N > 1 is not the N = 1 step plus an all-reduce
                       "step_structure": ("graph[fwd + loss + bwd + adamw]" if world == 1 else
                                          "graph[fwd + loss + bwd] + eager[grad all-reduce x3 buckets + adamw]") if graph
                       else "eager[fwd + loss + bwd (+ overlapped all-reduce) + optimizer.step]",
                       # float32 step: every matrix product runs in this build's kernels (no rocBLAS / hipBLASLt launch;
                       # the one library kernel left is MIOpen's 7 x 7 stem convolution); bf16 autocast: library GEMMs
                       # with their default selections
                       "library_gemms": "none" if args.dtype == "f32" else "rocBLAS / hipBLASLt defaults",
                       "final_loss": round(float(loss), 5)},
        }
        if not args.no_roofline:
            line.update(scan_rooflines(dev))
            line["roofline_conv"] = conv_roofline(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_size)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
