import os, sys, subprocess, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
if len(sys.argv) > 1:
    import openglottal_amd as og
    from openglottal_amd import synth
    feats = (32, 64, 128, 256)
    sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
    m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
    m.set_option("keep_taps", 1)
    fr = synth.random_gray_frames(64, seed=3)
    masks, areas, _ = m.segment(fr, want_mask=True)
    names = ["downs.0.b", "pool0", "downs.1.a", "downs.1.b", "downs.2.a", "downs.2.b", "downs.3.b", "bottleneck.b", "ups.0", "ups.1.a", "ups.1.b", "ups.3.b", "ups.5.b", "ups.6", "ups.7.a", "ups.7.b"]
    out = {}
    for n in names:
        try:
            out[n] = m.activation(n, 1)
        except Exception as e:
            print("skip", n, e)
    np.savez(sys.argv[1], areas=areas, **{k.replace(".", "_"): v for k, v in out.items()})
else:
    root = os.environ["GRAFT_REPO_ROOT"]
    for tag, lib in (("pk", "libopenglottal_hip.so"), ("sc", "libopenglottal_hip_noEPI.so")):
        env = dict(os.environ, OPENGLOTTAL_HIP_LIB=f"{root}/openglottal_amd/{lib}")
        subprocess.run([sys.executable, __file__, f"/tmp/{tag}.npz"], env=env, check=True)
    a, b = np.load("/tmp/pk.npz"), np.load("/tmp/sc.npz")
    print("areas equal", np.array_equal(a["areas"], b["areas"]))
    for k in a.files:
        if k == "areas": continue
        x, y = a[k], b[k]
        bad = np.argwhere(x != y)
        print(k, x.shape, "differing", len(bad), "max", float(np.abs(x - y).max()) if len(bad) else 0)
        if len(bad):
            print("   first:", bad[:6].tolist(), " channels:", np.unique(bad[:, -3] if x.ndim == 4 else bad[:, 0])[:20].tolist())
            ys = np.unique(bad[:, -2]); xs = np.unique(bad[:, -1])
            print("   rows:", ys[:24].tolist(), " cols:", xs[:24].tolist())
            break
