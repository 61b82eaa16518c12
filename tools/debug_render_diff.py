"""Debug helper (GPU box): prints pixels where the HIP pass and the oracle disagree."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scene, device
from oracle import oracle
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w, h = 160, 90
casa = scene.casa_scene('assets/casa_obj.glb'); sky = scene.procedural_sky(256, 128)
ctx = device.Context(0); sb = device.SceneBuffers(ctx, casa, sky); p = device.PathTracingPass(ctx, w, h); p.bind_scene(sb)
prm = scene.make_params(w, h, scene.casa_camera(), sky, spp=1, max_depth=depth)
p.set_params(prm); p.draw(); out = p.download()
ref, cnt = oracle.OracleScene(casa, sky).render(prm)
for k in ("directLum", "indirectLum"):
    a = out[k].reshape(-1, 3).astype(np.float64); b = ref[k].reshape(-1, 3).astype(np.float64)
    tol = 1e-3 * np.maximum(np.maximum(abs(a), abs(b)), 1e-4)
    bad = np.nonzero(~((abs(a - b) <= tol) | (a == b)).all(1))[0]
    print(k, 'bad', len(bad))
    inst = ref['instanceIndex'].reshape(-1)
    for i in bad[:40]:
        ii = int(inst[i]); mp = casa.instances['materialPtr'][ii] if ii >= 0 else -1
        rough = casa.materials[mp + 4] if ii >= 0 else -1
        print(i % w, i // w, 'inst', ii, 'rough', rough, 'gpu', a[i], 'cpu', b[i])
for k in out:
    a = out[k]; b = ref[k]
    eq = (a.view(np.uint32) == b.view(np.uint32)) | (a == b)
    print(k, 'bit-equal fraction', eq.mean(), 'max abs diff', np.nanmax(np.abs(a.astype(np.float64) - b)))
