import sys, torch, numpy as np
sys.path.insert(0, '.')
import streamyolo_amd as sy
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, load_bn_stats
dev = torch.device('cuda:0')
cfg = O.OracleConfig.named('l')
sd = synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats('l'))
x = synth_frames(1, 600, 960, seed=2)
ref = O.forward_eval({k: v.clone() for k, v in sd.items()}, x, cfg)
m = sy.build_model('l'); m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
for dt in ('fp32', 'fp16', 'bf16'):
    m.set_compute_dtype(dt)
    with torch.no_grad():
        out = m(x.to(dev)).cpu().float()
    print(dt, 'finite', bool(torch.isfinite(out).all()), 'ref max', float(ref.abs().max()))
    for nm, sl in (('xy', slice(0, 2)), ('wh', slice(2, 4)), ('obj', slice(4, 5)), ('cls', slice(5, 13))):
        d = (out[..., sl] - ref[..., sl]).abs()
        print('   %-3s max|d| %.4g  ref max %.4g  median|d| %.3g  rel %.3e' % (nm, float(d.max()), float(ref[..., sl].abs().max()), float(d.median()), float(d.max() / ref[..., sl].abs().max())))
    # log-space error of wh
    lw = (out[..., 2:4].clamp_min(1e-9).log() - ref[..., 2:4].clamp_min(1e-9).log()).abs()
    print('   log(wh) max err %.4g median %.3g' % (float(lw.max()), float(lw.median())))
