"""CPU: the error-budget emulator behind the mixed precision mode (oracle/error_budget.py).

The mixed mode's per-layer operand table (perceptor_amd/engine/adm_mixed.py: MIXED_SINGLE_STANDARD) was chosen on this emulator -- the
oracle's forward with one rounding point per tensor class the HIP engines round.  This test pins the decision without a GPU: on the shipped
558 M-parameter net at 128x128 the emulated f16 engine lands where the real one was measured (max-abs 2.6e-3), and the emulated mixed policy
(split storage, the table's single-operand layers, plain f16 blocks from 1/32 resolution down) stays under the contract's 1e-3 with margin.
The GPU side of the same statement is tests/test_gpu_mixed.py."""
import torch


def test_emulated_modes_on_the_shipped_net():
    from oracle import adm_unet as O, error_budget as E
    from perceptor_amd.engine.adm_mixed import MIXED_SINGLE_STANDARD
    from perceptor_amd.utils.synth import seeded_noise, synth_state_dict
    cfg = O.openimages_config()
    sd = {k: v.float() for k, v in synth_state_dict(O.state_dict_shapes(cfg), 0).items()}
    x, t = seeded_noise((1, 3, 128, 128), 1234), torch.tensor([500])
    ref = E.forward(sd, cfg, x, t, E.Policy("f16"))
    err = lambda P: (E.forward(sd, cfg, x, t, P) - ref)
    d16 = err(E.Policy("f16", E.CLASSES))
    assert 1.5e-3 < float(d16.abs().max()) < 4e-3 and 4e-4 < float(d16.pow(2).mean().sqrt()) < 8e-4      # GPU f16 engine: 2.6e-3 max-abs
    P = E.Policy("f16", split=("c1", "h", "sk", "x0", "at", "emb"))
    P.single, P.plain_from = set(MIXED_SINGLE_STANDARD), 32
    dm = err(P)
    rms, mx = float(dm.pow(2).mean().sqrt()), float(dm.abs().max())
    assert rms < 1.5e-4 and mx < 8e-4, (rms, mx)                        # GPU mixed engine: 2.4e-4 on the reference golden of this size
    # every layer name of the table exists in the net (a renamed block would silently become a doubled-operand layer)
    P0 = E.Policy("f16")
    E.forward(sd, cfg, x, t, P0)
    names = {n for n, _, _ in P0.seen}
    assert MIXED_SINGLE_STANDARD <= names and len(names) == 99
