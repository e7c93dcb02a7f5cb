import os as _os; _os.makedirs('/tmp/s360_fuzz', exist_ok=True)
import os, sys, json, subprocess
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
if len(sys.argv) > 2:
    from surround360_amd import _capi
    _capi.LIB_PATH = sys.argv[1]
    from surround360_amd import render as R
    import numpy as np, rigutil
    import oracle_lib as O
    kw = json.loads(sys.argv[2]); cam = kw.pop('cam')
    path = rigutil.scaled_rig_json(ROOT + '/tests/golden/rig_17cam.json', '/tmp/s360_fuzz/rig_tiny_%d.json' % cam, cam/2048.0)
    side, top, bottom = rigutil.frame_inputs(path, cam)
    rig = R.RigDescription(path)
    cams,_ = O.load_rig(path)
    of = O.Frame(cams, O.make_params(**kw))
    try:
        want,_ = of.render(side, top, bottom)
    except Exception as e:
        want = None; print('ORACLE FAILED', repr(e)[:150])
    try:
        ctx = R.Context(rig, R.make_params(**kw)); ctx.upload_frame(side, top, bottom); ctx.render(); got = ctx.download_equirect()
    except _capi.S360Error as e:
        got = None; print('LIB ERROR', str(e)[:150])
    if want is not None and got is not None:
        print('EQUAL' if (got.shape==want.shape and np.array_equal(got,want)) else 'DIFFER %s %s %d' % (got.shape, want.shape, int((got!=want).sum()) if got.shape==want.shape else -1))
    sys.exit(0)
cases=[]
for (w,h) in [(14,64),(28,7),(28,14),(42,21),(56,28),(70,33),(98,49),(126,63),(140,5),(14,300),(280,9)]:
    for poles in (0,1):
        cases.append(dict(cam=64, eqr_width=w, eqr_height=h, enable_top=poles, enable_bottom=poles, final_eqr_width=0, final_eqr_height=0, sharpening=0.25 if poles else 0.0, side_alpha_feather_size=7, std_alpha_feather_size=3))
cases.append(dict(side_alpha_feather_size=7, std_alpha_feather_size=5, cam=16, eqr_width=56, eqr_height=28, enable_top=1, enable_bottom=1, final_eqr_width=40, final_eqr_height=44, sharpening=0.25))
cases.append(dict(side_alpha_feather_size=4, std_alpha_feather_size=1, cam=8, eqr_width=28, eqr_height=14, enable_top=1, enable_bottom=1, final_eqr_width=3, final_eqr_height=2, sharpening=0.25))
cases.append(dict(side_alpha_feather_size=2, std_alpha_feather_size=3, cam=4, eqr_width=140, eqr_height=70, enable_top=1, enable_bottom=1, final_eqr_width=300, final_eqr_height=300, sharpening=0.0))
env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0:detect_stack_use_after_return=0')
if 'asan' in sys.argv[1]:
    import glob
    env['LD_PRELOAD']=subprocess.check_output(['/opt/rocm/lib/llvm/bin/clang++','-print-file-name=libclang_rt.asan-x86_64.so']).decode().strip()
for kw in cases:
    r = subprocess.run([sys.executable, __file__, sys.argv[1], json.dumps(kw)], capture_output=True, text=True, timeout=900, env=env)
    outl = [l for l in r.stdout.strip().splitlines() if l]
    errl = [l for l in r.stderr.strip().splitlines() if 'ERROR' in l or 'runtime error' in l or 'SUMMARY' in l]
    print({k:kw[k] for k in ('cam','eqr_width','eqr_height','enable_top','final_eqr_width','final_eqr_height')}, '->', 'rc', r.returncode, ' | '.join(outl[-2:]), ' | '.join(errl[:2])[:300], flush=True)
