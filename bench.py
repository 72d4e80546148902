#!/usr/bin/env python3
"""bench.py — denoising steps/s of the guided-diffusion sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
N > 1 without a torch.distributed.run environment: this process parses the arguments and, before it touches the GPU,
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (one rank per GPU over RCCL), relays
rank 0's JSON line and exits with the child's code.  Under torch.distributed.run (WORLD_SIZE set) it is a rank.

Workload (BASELINE.json metric "denoising steps/sec (UNet+CLIP-grad) at 512x512 batch 8"):
  config c5 (default) = configs[4] per-GPU share: GuidedDiffusion "standard" UNet @512x512,
  batch 8 per GPU (weak scaling: N GPUs sample 8N independent chains), OpenCLIP ViT-L/14
  guidance gradient, bf16 MFMA, synthetic weights/inputs.
  One step = UNet eps-prediction -> denoised images -> CLIP loss forward+backward to the image
  -> Predictions.guided -> Predictions.step (DDIM).

Prints ONE JSON line on rank 0 (contract in the task statement) incl. "roofline" and "cpu_baseline".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "precise": 2500.0, "mixed": 2500.0}   # precise = f16 MFMA on hi + lo pairs: twice the MFMA work per algorithmic FLOP   # dense MFMA peak, MI355X_MICROARCH.md (never the 2:1-sparse figure)

# forward GFLOP / sample (torch flop counter on the reference modules)
UNET_GFLOP = {"standard": {512: 3964.7, 256: 989.0}, "pixelart": {256: 497.5, 64: 497.5 / 16},
              "yfcc_2": {512: 2314.3, 256: 578.4}, "yfcc_1": {512: 2058.1}, "cc12m_1_cfg": {256: 831.8}}
CLIP_FWD_GFLOP = {"ViT-B-32": 8.82, "ViT-B-16": 35.13, "ViT-L-14": 162.03, "ViT-H-14": 334.59}

CONFIGS = {
    # name: (model, resolution, batch per GPU, clip arch)
    "c5": ("standard", 512, 8, "ViT-L-14"),
    "c1": ("cc12m_1_cfg", 256, 1, None),            # configs[0]: the reference's CPU-runnable case, here on the GPU
    "c2": ("standard", 256, 4, None),
    "c3": ("yfcc_2", 512, 8, "ViT-B-32"),
    "c5-noclip": ("standard", 512, 8, None),
    "c4": ("stable-diffusion-v1", 512, 4, None),      # configs[3]: SD-v1 latent UNet 512x512 with CFG, batch 32 over 8 GPUs = 4 per GPU
    "smoke": ("pixelart", 64, 2, None),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="c5", choices=sorted(CONFIGS))
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "mixed", "precise"],
                   help="UNet arithmetic: bf16 / f16 single-pass MFMA; mixed (GD UNets: hi + lo storage, doubled operands on the layers the error "
                        "budget names: eps max-abs error < 1e-3 vs the fp32 reference at ~1.4x the f16 MFMA work); precise (hi + lo everywhere, 2x)")
    p.add_argument("--no-modes", action="store_true", help="skip the short timing + parity of the other precision modes after the timed region (\"modes\" in the line)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--opt", default="", help="A/B switches of the library: comma-separated key=value pairs for pmi_set_option (e.g. 10=0: no conv split-K)")
    p.add_argument("--sd-bf16", action="store_true", help="config c4: bf16 operands instead of the f16 the reference runs SD in")
    p.add_argument("--no-kernel-events", action="store_true")
    p.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (helps launch-bound small configs such as c1)")
    p.add_argument("--dump-kernels", default=None, help="write per-launch (ms, GFLOP, MB) of the timed conv3x3 launches of the last step to this file")
    p.add_argument("--backward", action="store_true", help="row f2: time UNet forward (training mode, tape kept) + input-gradient backward of a fixed "
                                                            "output gradient instead of the sampling step (no CLIP leg, no update)")
    p.add_argument("--rehearse", action="store_true", help="launcher / collective rehearsal without the model: every rank joins the process group "
                                                            "(gloo when there is no GPU), all-gathers a small tensor and rank 0 prints a JSON line")
    return p.parse_args()


def self_launch(a) -> int:
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as fresh child processes (this parent has made no GPU call),
    stream their output through, return the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this host driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rehearse(a, rank, local_rank, world):
    """Process-group plumbing only: init, all_gather, barrier, max-reduced time, JSON line on rank 0."""
    import torch.distributed as dist
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() > local_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if use_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.perf_counter()
    mine = torch.full((4, 3, 8, 8), float(rank), device=dev)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    dist.barrier()
    ok = all(float(g.mean()) == float(r) for r, g in enumerate(got))
    tmax = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "backend": "nccl" if use_gpu else "gloo", "rccl_ranks": world if use_gpu else 0,
                          "all_gather_ok": ok, "seconds": round(float(tmax.item()), 4)}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


def cpu_baseline_v(model_name, res):
    """v-diffusion nets: oracle forward at batch 1, 64x64 (cc12m_1) or 128x128 (yfcc), scaled by pixels x batch."""
    from oracle import vdiff as ov
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    spec = {"yfcc_2": ov.yfcc2_spec, "yfcc_1": ov.yfcc1_spec, "cc12m_1_cfg": ov.cc12m1_spec}[model_name]()
    sd = synth_state_dict(ov.state_dict_shapes(spec), 0)
    sres = 128 if res >= 512 else 64
    x = seeded_noise((1, 3, sres, sres), 1234)
    t = torch.tensor([0.5])
    ce = seeded_noise((1, 512), 11) if spec["cond"] else None
    ov.vdiff_forward(sd, spec, x, t, ce)
    t0 = time.time()
    ov.vdiff_forward(sd, spec, x, t, ce)
    return time.time() - t0, sres, torch.get_num_threads()


def cpu_baseline(model_name, res, nb, clip_arch, clip_loss, hip_model, dev, seed=0):
    """Checker leg (rank 0, N = 1, outside the timed region): ONE full denoising step of the oracle -- the CPU fp32 restatement of the
    reference path -- timed on the host cores at batch 1 and min(res, 256)^2: UNet -> denoised -> resize -> ViT forward + backward to the
    image -> guided -> DDIM step.  The UNet and the per-pixel updates scale with batch x pixels, the CLIP leg (always 224^2) with batch.
    The same sample is run through the HIP model: max |eps_hip - eps_oracle| is the parity figure of the timed mode."""
    from oracle import adm_unet, clip_vit, sampling
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    cores = torch.get_num_threads()
    cfg = adm_unet.openimages_config() if model_name == "standard" else adm_unet.pixelart_config()
    sd = synth_state_dict(adm_unet.state_dict_shapes(cfg), seed)
    sres = min(res, 256)
    x = seeded_noise((1, 3, sres, sres), 1234)
    images = x * 0.5 + 0.5
    fi, ti = torch.tensor([600]), torch.tensor([550])
    al, sg = sampling.gd_tables()
    vcfg = clip_vit.VIT_CONFIGS[clip_arch] if clip_arch else None
    vsd = synth_state_dict(clip_vit.vit_state_dict_shapes(vcfg), 0) if clip_arch else None
    targets = clip_loss.encodings.detach().cpu().float() if clip_arch else None

    def step():
        t0 = time.time()
        eps = adm_unet.adm_unet_forward(sd, cfg, x, fi)[:, :3]
        t1 = time.time()
        den = (sampling.eps_denoised_xs(images, eps, al[fi], sg[fi]) + 1) / 2
        if clip_arch:
            _, grad = clip_vit.clip_loss_and_grad(vsd, vcfg, den, targets, torch.ones(len(targets)), clip_loss.model.quick_gelu)
            eps_g = sampling.guided(eps, grad, sg[fi], 0.5, 1e-6)
        else:
            eps_g = eps
        t2 = time.time()
        nxt = sampling.eps_step(images, eps_g, al[fi], sg[fi], al[ti], sg[ti])
        t3 = time.time()
        return eps, nxt, (t1 - t0, t2 - t1, t3 - t2)

    step()                                                     # warm-up (allocator, thread pool)
    eps_ref, _, (t_unet, t_clip, t_upd) = step()
    scale_px = (res / sres) ** 2 * nb
    sec_step = t_unet * scale_px + t_clip * nb + t_upd * scale_px
    parity = None
    if hip_model is not None:
        eps_hip = hip_model.predicted_noise(images.to(dev), fi.to(dev)).float().cpu()
        parity = {"eps_max_abs_err": float((eps_hip - eps_ref).abs().max()), "eps_max_abs": float(eps_ref.abs().max()),
                  "sample": f"batch 1 @{sres}x{sres}, index 600, vs the CPU fp32 oracle on the same weights and input"}
    sample = (f"one full oracle step at batch 1 @{sres}x{sres}: UNet {t_unet:.2f}s + CLIP {clip_arch or 'none'} fwd+bwd and guidance {t_clip:.2f}s + update {t_upd:.3f}s; "
              f"UNet/update scaled x{scale_px:.0f} (batch x pixels), CLIP leg x{nb} (batch)")
    return sec_step, sample, cores, parity


def precision_modes(a, out, model_name, res, eager_step, images, dev, steps=6):
    """After the timed region (rank 0, N = 1): the same step in the other precision modes of the UNet, `steps` steps each after 2 of warm-up,
    HIP-event time on the launch stream, and the eps parity of each mode on the cpu_baseline's sample (batch 1 @256x256 vs the CPU fp32
    oracle).  The timed mode's own entry repeats the line's figures.  north_star's bar is eps max-abs error < 1e-3."""
    from oracle import adm_unet
    from perceptor_amd import models
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    modes = {a.dtype: {"ms_per_step": out["ms_per_step"], "eps_max_abs_err": (out.get("parity") or {}).get("eps_max_abs_err"), "timed": True}}
    cfg = adm_unet.openimages_config() if model_name == "standard" else adm_unet.pixelart_config()
    sres = min(res, 256)
    x = seeded_noise((1, 3, sres, sres), 1234)
    fi = torch.tensor([600])
    eps_ref = None
    for name in ("bf16", "f16", "mixed"):
        if name in modes:
            continue
        m = models.GuidedDiffusion(model_name, dtype=name).to(dev)
        img = images.clone()
        for i in range(2):
            img = eager_step(img, i, m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            img = eager_step(img, 2 + i, m)
        e1.record()
        torch.cuda.synchronize()
        entry = {"ms_per_step": round(e0.elapsed_time(e1) / steps, 3), "steps": steps, "outputs_finite": bool(torch.isfinite(img).all().item())}
        if eps_ref is None:
            sd = synth_state_dict(adm_unet.state_dict_shapes(cfg), 0)
            eps_ref = adm_unet.adm_unet_forward(sd, cfg, x, fi)[:, :3]
        eps = m.predicted_noise((x * 0.5 + 0.5).to(dev), fi.to(dev)).float().cpu()
        entry["eps_max_abs_err"] = float((eps - eps_ref).abs().max())
        modes[name] = entry
        del m
        torch.cuda.empty_cache()
    return modes


def main_sd(a, rank, world, dev, dist):
    """config c4 (BASELINE configs[3]): one step = the unconditioned + conditioned SD-v1 UNet evaluation of a batch of
    4 latents (one batched launch sequence, batch 8), classifier-free guidance and the DDIM update.  Prompt encodings are computed
    once before the timed region, as in the reference's sample() (stable_diffusion.py:415-426)."""
    from perceptor_amd import models
    from perceptor_amd.engine import sd as sd_engine
    from perceptor_amd.utils.synth import seeded_noise
    _, res, nb, _ = CONFIGS["c4"]
    dtype = "f16" if a.dtype == "bf16" and not a.sd_bf16 else a.dtype        # the reference runs SD in fp16 (fp16=True default)
    model = models.StableDiffusion(fp16=(dtype == "f16")).to(dev)
    ids = torch.full((2, 77), 49407, dtype=torch.int64)
    ids[:, 0] = 49406
    ids[1, 1:9] = torch.tensor([1125, 539, 320, 2368, 525, 320, 4558, 267])      # a fixed 8-token prompt; row 0 is the empty prompt
    neutral, positive = model.conditioning(token_ids=ids[:1]), model.conditioning(token_ids=ids[1:])
    lat = seeded_noise((nb * world, 4, res // 8, res // 8), 1234)[rank * nb:(rank + 1) * nb].to(dev)
    sched = model.schedule_indices(n_steps=max(a.steps + a.warmup + 1, 50))

    def one_step(lat, i):
        fi, ti = sched[i % len(sched)]
        un, pos = model.predictions_pair(lat, fi, neutral, positive)
        return un.classifier_free_guidance(pos, guidance_scale=7.0).step(ti)

    if a.graph:
        from perceptor_amd.engine.graph import GraphedStep

        def step_fn(x, fi, ti):
            un, pos = model.predictions_pair(x, fi, neutral, positive)
            return un.classifier_free_guidance(pos, guidance_scale=7.0).step(ti)
        gstep = GraphedStep(step_fn, lat, sched[0][0], sched[0][1])

        def one_step(lat, i):   # noqa: F811
            fi, ti = sched[i % len(sched)]
            return gstep(lat, fi, ti).clone()
    for i in range(a.warmup):
        lat = one_step(lat, i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        lat = one_step(lat, a.warmup + i)
        ev[i][1].record()
    if dist is not None:
        gathered = [torch.empty_like(lat) for _ in range(world)]
        dist.all_gather(gathered, lat)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank != 0:
        return
    step_ms_dev = sum(s.elapsed_time(e) for s, e in ev) / a.steps
    gflop_eval = sd_engine.unet_gflop(sd_engine.SD_V1, res // 8, res // 8, 77)
    tflop_step = 2 * nb * gflop_eval / 1e3
    achieved = tflop_step / (step_ms_dev / 1e3)
    peak = PEAK_TFLOPS["f16" if dtype == "f16" else "bf16"]
    out = {
        "metric": "denoising steps/sec (c4)", "value": round(a.steps / elapsed * world, 4),
        "unit": f"steps/s (batch-{nb} CFG steps summed over GPUs)", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"StableDiffusion v1 latent UNet {res}x{res} (64x64 latents), batch {nb}/GPU, classifier-free guidance "
                               "(unconditioned + conditioned evaluation = UNet batch 8) + DDIM eta=0, 77-token CLIP ViT-L/14 text context, synthetic weights",
                   "name": "c4", "global_batch": nb * world, "parallelism": f"replica-sharded chains x{world}"},
        "outputs_finite": bool(torch.isfinite(lat).all().item()), "rccl_ranks": world if dist is not None else 0,
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": None, "kernel": "whole step (no per-kernel events on this config)",
                     "algorithmic_gflop_per_unet_eval_per_sample": round(gflop_eval, 1)},
    }
    if not a.no_cpu_baseline and world == 1:
        from oracle import sd as osd
        sres = 16
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        usd = {k: v.detach().float().cpu() for k, v in model.unet.state_dict().items()}
        x = seeded_noise((1, 4, sres, sres), 5)
        ctx = positive.encodings.float().cpu()
        t0 = time.perf_counter()
        with torch.no_grad():
            want = osd.unet_forward(usd, osd.SD_V1, x, torch.tensor([600]), ctx)
        t_eval = time.perf_counter() - t0
        got = model.predicted_noise(x.to(dev), 600, positive).cpu()
        scale = (res // 8 / sres) ** 2 * nb * 2
        out["cpu_baseline"] = {"value": round(1.0 / (t_eval * scale), 6), "unit": f"steps/s (batch-{nb} CFG steps)", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"one oracle UNet evaluation at batch 1, {sres}x{sres} latents = {t_eval:.2f}s, scaled x{scale:.0f} "
                                                         "(pixels x batch x 2 evaluations per CFG step)"}
        out["parity"] = {"eps_max_abs_err": float((got - want).abs().max()), "eps_max_abs": float(want.abs().max()),
                         "sample": f"batch 1, {sres}x{sres} latents, index 600, vs the CPU fp32 oracle (pinned on the reference's vendored CompVis UNet, tests/golden/sd_ldm_*.npz)"}
    print(json.dumps(out), flush=True)


def conv_kernel_name(desc):
    """Kernel a timed 3x3-convolution launch ran in, from its tile config (csrc/conv3x3.hip: pmi_conv3x3_halo_config)."""
    cfg = int(desc.split(" cfg")[1].split()[0])
    return "conv3x3_wd_kernel" if cfg >= 4 else "conv3x3_halo_kernel"


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))                 # parent: no GPU call made, children are fresh processes
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if a.rehearse:
        raise SystemExit(rehearse(a, rank, local_rank, world))
    if a.opt:
        from perceptor_amd import _hip
        for kv in a.opt.split(","):
            k, v = kv.split("=")
            _hip.lib().pmi_set_option(int(k), int(v))
    torch.cuda.set_device(local_rank)            # before the process group: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.config == "c4":
        main_sd(a, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    from perceptor_amd import models
    from perceptor_amd.engine import ops
    from perceptor_amd.utils.synth import seeded_noise

    model_name, res, nb, clip_arch = CONFIGS[a.config]
    if a.backward:
        clip_arch = None
    is_v = model_name not in ("standard", "pixelart")
    model = (models.VelocityDiffusion(model_name, dtype=a.dtype) if is_v else models.GuidedDiffusion(model_name, dtype=a.dtype)).to(dev)
    cond = seeded_noise((1, 1, 512), 11).to(dev) if model_name.startswith("cc12m") else None
    clip_loss = None
    if clip_arch is not None:
        from perceptor_amd import losses
        clip_loss = losses.OpenCLIP(clip_arch, "synthetic", dtype="bf16").to(dev)
        tg = torch.nn.functional.normalize(seeded_noise((2, clip_loss.model.output_dim), 7))
        clip_loss.add_encodings_(tg.to(dev))
    # one global noise batch, sliced per rank: results do not depend on the number of GPUs
    noise = seeded_noise((nb * world, 3, res, res), 1234)[rank * nb:(rank + 1) * nb]
    images = (noise * 0.5 + 0.5).to(dev)
    n_sched = max(a.steps + a.warmup + 1, 50)
    sched = model.schedule_ts(n_steps=n_sched).to(dev) if is_v else model.schedule_indices(n_steps=n_sched, rho=7.0)

    def one_step(images, i, model=model):
        fi, ti = sched[i % len(sched)]
        pred = model.predictions(images, fi, cond) if is_v else model.predictions(images, fi)
        if clip_loss is not None:
            _, grad = clip_loss.loss_and_grad(pred.denoised_images, n_total=nb * world)
            pred = pred.guided(grad, guidance_scale=0.5, clamp_value=1e-6)
        return pred.step(ti)
    eager_step = one_step
    tape_gb = None
    if a.backward:
        # what a script differentiating through the UNet pays per evaluation (losses/velocity_diffusion.py:33-61 guided_resample_; upstream
        # GuidedDiffusion.predicted_noise under autograd, guided_diffusion.py:125-133): training-mode forward + backward to the input
        a.no_kernel_events, a.no_modes = True, True
        probe = seeded_noise((nb, 3, res, res), 99).to(dev)
        sdict = model.model.state_dict()

        def tensors(o, seen):
            if torch.is_tensor(o):
                if o.data_ptr() not in seen:
                    seen[o.data_ptr()] = o.numel() * o.element_size()
            elif isinstance(o, (list, tuple)):
                for v in o:
                    tensors(v, seen)
            elif isinstance(o, dict):
                for v in o.values():
                    tensors(v, seen)

        def one_step(images, i, model=model):   # noqa: F811
            nonlocal tape_gb
            fi, _ = sched[i % len(sched)]
            if is_v:
                if cond is not None:
                    raise SystemExit("--backward: unconditional nets only (c2, c3, c5-noclip)")
                _, tape = model.engine.forward_train(images, fi.reshape(-1).expand(nb), None)
            else:
                _, tape = model.engine.forward_train(images, model.indices(fi).expand(nb), sdict, out_channels=3)
            if tape_gb is None:
                seen = {}
                tensors(tape, seen)
                tape_gb = sum(seen.values()) / 1e9
            g = model.engine.backward(tape, probe, sdict)
            return images

    if a.graph:
        from perceptor_amd.engine.graph import GraphedStep
        a.no_kernel_events = True            # per-launch events cannot be recorded inside a replayed graph

        def step_fn(img, fi, ti):
            pred = model.predictions(img, fi, cond) if is_v else model.predictions(img, fi)
            if clip_loss is not None:
                _, grad = clip_loss.loss_and_grad(pred.denoised_images, n_total=nb * world)
                pred = pred.guided(grad, guidance_scale=0.5, clamp_value=1e-6)
            return pred.step(ti)
        gstep = GraphedStep(step_fn, images, sched[0][0], sched[0][1])

        def one_step(images, i):   # noqa: F811  (the captured step replaces the eager one)
            fi, ti = sched[i % len(sched)]
            return gstep(images, fi, ti).clone()
    for i in range(a.warmup):
        images = one_step(images, i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if not a.no_kernel_events:
        ops.KERNEL_EVENTS = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        images = one_step(images, a.warmup + i)
        ev[i][1].record()
    if dist is not None:
        gathered = [torch.empty_like(images) for _ in range(world)]
        dist.all_gather(gathered, images)       # RCCL over xGMI: the only collective of the job
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    rank_ms = [elapsed / a.steps * 1e3]
    if dist is not None:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / a.steps * 1e3 for t in every]
        tmax = mine.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
    finite = bool(torch.isfinite(images).all().item())
    checksum = float(images.double().abs().sum().item())     # same schedule and seed -> comparable between --graph / --chains / eager runs

    if rank == 0:
        step_ms_dev = sum(s.elapsed_time(e) for s, e in ev) / a.steps
        gflop_sample = UNET_GFLOP[model_name][res] * (2.0 if a.backward else 1.0) + (2 * CLIP_FWD_GFLOP[clip_arch] if clip_arch else 0.0)
        tflop_step = gflop_sample * nb / 1e3
        achieved = tflop_step / (step_ms_dev / 1e3)
        step_roof = {"achieved": round(achieved, 2), "frac": round(achieved / PEAK_TFLOPS[a.dtype], 4),
                     "scope": "whole step (UNet fwd + CLIP fwd+bwd + update): algorithmic FLOP / HIP-event step time"}
        roof = {"bound": "mfma", "achieved": None, "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s", "frac": None, "traffic": None,
                "step": step_roof}
        if kernel_events:
            # dominant kernels: every 3x3-convolution launch of the timed steps (conv3x3_wd_kernel for the 256-multiple Cout
            # layers, conv3x3_halo_kernel for the rest), HIP events on the launch stream
            def agg(evs):
                ms = sum(e[0].elapsed_time(e[1]) for e in evs)
                fl = sum(e[2] for e in evs)
                by = sum(e[3] for e in evs)
                t = fl / 1e12 / (ms / 1e3)
                return {"achieved": round(t, 2), "frac": round(t / PEAK_TFLOPS[a.dtype], 4), "launches": len(evs),
                        "avg_ms": round(ms / len(evs), 4), "share_of_step": round(ms / a.steps / step_ms_dev, 3),
                        "algorithmic_gflop_per_launch": round(fl / 1e9 / len(evs), 1),
                        "algorithmic_bytes_per_launch": round(by / len(evs)),
                        "algorithmic_GBps": round(by / 1e9 / (ms / 1e3), 1)}
            by_kernel = {}
            for e in kernel_events:
                by_kernel.setdefault(conv_kernel_name(e[4]), []).append(e)
            roof.update({"kernel": "3x3 convolution as implicit GEMM on MFMA: conv3x3_wd_kernel (csrc/conv_wd.hip) + "
                                   "conv3x3_halo_kernel (csrc/conv3x3.hip), all launches of the timed steps"})
            roof.update(agg(kernel_events))
            roof["by_kernel"] = {k: agg(v) for k, v in sorted(by_kernel.items())}
            if a.dump_kernels:
                per = len(kernel_events) // a.steps
                with open(a.dump_kernels, "w") as f:
                    for k in range(per):
                        evs = [kernel_events[st * per + k] for st in range(a.steps)]
                        ms = sum(e[0].elapsed_time(e[1]) for e in evs) / a.steps
                        f.write(f"{k:3d} {ms:8.4f} ms {evs[0][2] / 1e9:10.1f} GFLOP {evs[0][3] / 1e6:9.1f} MB {evs[0][2] / 1e9 / ms:8.1f} TFLOP/s {evs[0][3] / 1e6 / ms:8.1f} GB/s {evs[0][4]}\n")
            # HBM bytes per conv3x3 launch from the committed PMC passes of this command -- only while the profile belongs to these kernels
            # (sha of the conv sources recorded by tools/pmc_summary.py); a stale profile leaves `traffic` null and says so
            pmc = os.path.join(ROOT, "profiles", "r03_pmc_hbm.json" if a.dtype == "bf16" else f"r03_pmc_hbm_{a.dtype}.json")
            if a.config == "c5" and os.path.exists(pmc):
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from pmc_summary import kernel_src_sha16
                with open(pmc) as f:
                    rows = json.load(f)
                if rows.get("_kernel_src_sha16") == kernel_src_sha16():
                    krows = [rows[k] for k in ("conv3x3_wd_kernel", "conv3x3_halo_kernel") if k in rows]
                    if krows:
                        roof["traffic"] = round(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in krows) / sum(r["launches"] for r in krows))
                        roof["traffic_source"] = (f"profiles/{os.path.basename(pmc)} (kernel sources unchanged since its collection): (2*FETCH_SIZE + WRITE_SIZE) KiB per "
                                                  "launch, launch-weighted over conv3x3_wd_kernel and conv3x3_halo_kernel; separate rocprofv3 --pmc passes of this command")
                else:
                    roof["traffic_stale_profile"] = f"profiles/{os.path.basename(pmc)} was collected on other conv kernel sources: traffic left null"
        else:
            roof.update({"achieved": step_roof["achieved"], "frac": step_roof["frac"], "kernel": "whole step (no per-kernel events)"})
        out = {
            "metric": (f"UNet forward + input-gradient evaluations/sec ({a.config})" if a.backward else
                       "denoising steps/sec (UNet+CLIP-grad) at 512x512 batch 8" if a.config == "c5" else f"denoising steps/sec ({a.config})"),
            "value": round(a.steps / elapsed * world, 4),
            "unit": f"steps/s (batch-{nb} steps summed over GPUs)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{'VelocityDiffusion' if is_v else 'GuidedDiffusion'} '{model_name}' UNet {res}x{res}, batch {nb}/GPU"
                                   + (f" + OpenCLIP {clip_arch} guidance (fwd+bwd to image)" if clip_arch else " (no CLIP)")
                                   + ", DDIM eta=0, synthetic weights", "name": a.config,
                       "global_batch": nb * world, "parallelism": f"replica-sharded chains x{world}"},
            "outputs_finite": finite, "images_abs_sum": round(checksum, 3),
            "rccl_ranks": dist.get_world_size() if dist is not None else 0,
            "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},     # skew between the ranks' own clocks
            **({"tape_gb": round(tape_gb, 2), "flop_note": "roofline.step counts 2x the forward FLOP (forward + dX of every layer; no weight gradients)"} if a.backward else {}),
            "roofline": roof,
        }
        if dist is not None:
            assert out["rccl_ranks"] == a.gpus, (out["rccl_ranks"], a.gpus)
        if not a.no_cpu_baseline and world == 1:
            if is_v:
                t_unet, sres, cores = cpu_baseline_v(model_name, res)
                scale = (res / sres) ** 2 * nb
                sec_step = t_unet * scale * (gflop_sample / UNET_GFLOP[model_name][res])
                sample = (f"oracle UNet fwd batch 1 @{sres}x{sres} = {t_unet:.2f}s, scaled x{scale:.0f} by pixel*batch"
                          f" and x{gflop_sample / UNET_GFLOP[model_name][res]:.3f} for the CLIP FLOP share")
                parity = None
            else:
                sec_step, sample, cores, parity = cpu_baseline(model_name, res, nb, clip_arch, clip_loss, model, dev)
            out["cpu_baseline"] = {"value": round(1.0 / sec_step, 6), "unit": f"steps/s (batch-{nb} steps)", "cores": cores, "kind": "port",
                                   "sample": sample}
            if parity is not None:
                out["parity"] = parity
        if not a.no_modes and world == 1 and not is_v and a.config in ("c5", "c2", "c5-noclip"):
            out["modes"] = precision_modes(a, out, model_name, res, eager_step, images, dev)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
