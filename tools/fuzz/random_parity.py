"""Whole frames with random flags (eqr sizes, camera sizes, poles, final resize, sharpening, feathers, eye distance, flow\nalgorithm, one or two chained frames) on an emulated build of the library against the oracle, byte for byte.\nusage: python tools/fuzz/random_parity.py <libs360 build> drive <seed> <cases>"""
import os as _os; _os.makedirs('/tmp/s360_fuzz', exist_ok=True)
import os, sys, json, subprocess, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
if len(sys.argv) > 2 and sys.argv[2] != 'drive':
    from surround360_amd import _capi
    _capi.LIB_PATH = sys.argv[1]
    from surround360_amd import render as R
    import numpy as np, rigutil
    import oracle_lib as O
    kw = json.loads(sys.argv[2]); cam = kw.pop('cam'); cube = kw.pop('cube'); frames = kw.pop('frames')
    path = rigutil.scaled_rig_json(ROOT + '/tests/golden/rig_17cam.json', '/tmp/s360_fuzz/rig_rand_%d.json' % cam, cam/2048.0)
    rig = R.RigDescription(path)
    cams,_ = O.load_rig(path)
    okw = {k: v for k, v in kw.items() if not k.endswith('_alg')}
    okw['side_flow_search20'] = int(kw['side_flow_alg'] == 'pixflow_search_20')
    of = O.Frame(cams, O.make_params(**okw))
    ctx = R.Context(rig, R.make_params(**kw))
    res = []
    for f in range(frames):
        side, top, bottom = rigutil.frame_inputs(path, cam, yaw_deg=0.7*f, world_h=256)
        want,_ = of.render(side, top, bottom, use_prev=(f>0))
        ctx.upload_frame(side, top, bottom); ctx.render(f>0); got = ctx.download_equirect()
        res.append(bool(got.shape==want.shape and np.array_equal(got,want)))
    print('EQUAL' if all(res) else 'DIFFER %s' % res)
    sys.exit(0)
random.seed(int(sys.argv[3]) if len(sys.argv)>3 else 1)
n = int(sys.argv[4]) if len(sys.argv)>4 else 30
bad = 0
for i in range(n):
    cam = random.choice([32, 48, 64, 96, 128])
    w = 14 * random.randint(2, 40); h = random.randint(14, 200)
    poles = random.choice([0,1,1])
    fw = random.choice([0, 0, random.randint(8, 400)]); fh = 0 if fw == 0 else 2*random.randint(4, 200)
    kw = dict(cam=cam, cube=0, frames=random.choice([1,1,2]), eqr_width=w, eqr_height=h, enable_top=poles, enable_bottom=random.choice([poles, 0]) if poles else 0,
              final_eqr_width=fw, final_eqr_height=fh, sharpening=random.choice([0.0, 0.25, 0.9]),
              side_alpha_feather_size=random.randint(0, cam//2), std_alpha_feather_size=random.choice([1,3,5,9,15,31]),
              interpupilary_dist=random.choice([6.4, 3.0, 0.0]), zero_parallax_dist=random.choice([10000.0, 300.0]),
              side_flow_alg=random.choice(["pixflow_low","pixflow_low","pixflow_search_20"]), polar_flow_alg="pixflow_low")
    try:
        r = subprocess.run([sys.executable, __file__, sys.argv[1], json.dumps(kw)], capture_output=True, text=True, timeout=900)
        outl = [l for l in r.stdout.strip().splitlines() if l]
        status = outl[-1] if outl else 'rc %d %s' % (r.returncode, (r.stderr.strip().splitlines() or [''])[-1][:200])
    except subprocess.TimeoutExpired:
        status = 'TIMEOUT'
    if status != 'EQUAL': bad += 1
    if status != 'EQUAL': print(i, status[:120], json.dumps({k: kw[k] for k in kw if k not in ('polar_flow_alg','cube')}), flush=True)
print('done, not equal:', bad)
