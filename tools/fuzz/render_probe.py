import os as _os; _os.makedirs('/tmp/s360_fuzz', exist_ok=True)
import os, sys, subprocess, json
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
if len(sys.argv) > 2:
    from surround360_amd import _capi
    _capi.LIB_PATH = sys.argv[1]
    from surround360_amd import render as R, synth
    import numpy as np, rigutil
    kw = json.loads(sys.argv[2])
    for k,v in list(kw.items()):
        if v == "nan": kw[k] = float('nan')
        if v == "inf": kw[k] = float('inf')
    rig_path = rigutil.scaled_rig_json(ROOT + '/tests/golden/rig_17cam.json', '/tmp/s360_fuzz/rig_probe.json', 96/2048.0)
    rig = R.RigDescription(rig_path)
    side, top, bottom = synth.rig_frame(rig_path, 96, world_h=256, seed=1)
    try:
        ctx = R.Context(rig, R.make_params(**kw))
        ctx.upload_frame(side, top, bottom)
        ctx.render(False)
        eq = ctx.download_equirect()
        print("RENDERED", eq.shape)
    except _capi.S360Error as e:
        print("S360Error", str(e)[:160])
    sys.exit(0)
cases = [
 dict(eqr_width=-14, eqr_height=64), dict(eqr_width=28, eqr_height=0), dict(eqr_width=28, eqr_height=-5), dict(eqr_width=28, eqr_height=2),
 dict(eqr_width=28, eqr_height=7), dict(eqr_width=0, eqr_height=64), dict(eqr_width=14, eqr_height=64), dict(eqr_width=140, eqr_height=64, final_eqr_width=0, final_eqr_height=0),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=-1, final_eqr_height=64), dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=-1),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=1), dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, side_alpha_feather_size=-3),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, side_alpha_feather_size=100000),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, side_alpha_feather_size=0),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, std_alpha_feather_size=0),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, std_alpha_feather_size=-1),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, std_alpha_feather_size=2),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, std_alpha_feather_size=1001),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, interpupilary_dist="nan"),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, zero_parallax_dist=0.0),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, sharpening="nan"),
 dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=64, enable_top=1, enable_bottom=1, sharpening=-1.0),
 dict(eqr_width=1400000, eqr_height=64), dict(eqr_width=140, eqr_height=2000000000),
]
for kw in cases:
    r = subprocess.run([sys.executable, __file__, sys.argv[1], json.dumps(kw)], capture_output=True, text=True, timeout=600)
    last = (r.stdout.strip().splitlines() or ['<no stdout>'])[-1]
    print(kw, '->', 'rc', r.returncode, last, (r.stderr.strip().splitlines() or [''])[-1][:200] if r.returncode else '', flush=True)
