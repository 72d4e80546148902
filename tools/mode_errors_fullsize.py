import sys, torch
sys.path.insert(0, '.')
from perceptor_amd.engine import adm, adm_mixed
from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
cfg = adm.openimages_config(); sd = synth_state_dict(adm.state_dict_shapes(cfg), 0)
x = seeded_noise((8, 3, 512, 512), 1234); img = ((x + 1) / 2).cuda()
for tt in (600, 100, 950):
    t = torch.full((8,), tt).cuda()
    ref = adm.AdmEngine(cfg, sd, "cuda:0", "precise").forward(img, t, out_channels=3).cpu()
    torch.cuda.empty_cache()
    for mode in ("mixed", "f16", "bf16"):
        eng = adm_mixed.AdmMixedEngine(cfg, sd, "cuda:0") if mode == "mixed" else adm.AdmEngine(cfg, sd, "cuda:0", mode)
        y = eng.forward(img, t, out_channels=3).cpu()
        d = y - ref
        print(f"512x512 x8 t={tt} {mode}: max-abs {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} scale {float(ref.abs().max()):.3f}", flush=True)
        del eng; torch.cuda.empty_cache()
