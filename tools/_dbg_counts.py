import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import _harness as H
from _cases import random_pt_scene
import srt_amd
bad=0
for seed in list(range(300, 340)) + list(range(100000, 100016)) + list(range(200000, 200010)):
    scene, w, h, depth, use_bvh, spp = random_pt_scene(seed)
    try:
        o = H.OraclePT(scene, w, h, depth, use_bvh)
        cnt = np.zeros(8, np.uint64)
        want = o.epoch(seed, 3, spp, counters=cnt)
    except AssertionError:
        continue
    pt = srt_amd.Pathtracer(0); pt.set_params(w,h,1,depth,use_bvh); pt.build_scene(scene); pt.set_camera(scene["camera"])
    res={}
    for mode, elide in ((2, False), (4, True), (6, False), (6, True), (7, False), (7, True)):
        pt.set_kernel(mode); pt.set_elision(elide); pt.ray_count(reset=True)
        try:
            img = pt.render_epoch(seed, 3, spp)
        except srt_amd.SrtError as e:
            continue
        ok = np.array_equal(img.view(np.uint32), want.view(np.uint32)) or bool(((img.view(np.uint32)==want.view(np.uint32))|(np.isnan(img)&np.isnan(want))).all())
        res[(mode,elide)]=(pt.ray_count()[0], ok)
    vals=set(v[0] for v in res.values())
    if len(vals)>1 or not all(v[1] for v in res.values()):
        bad+=1
        print(seed, 'oracle rays', int(cnt[0]), res, 'lights', len(scene.get('lights',[])), 'env', scene.get('env',{}).get('type'), 'w,h,spp,depth', w,h,spp,depth)
    pt.close()
print('bad', bad)
