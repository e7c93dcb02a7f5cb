import os as _os; _os.makedirs('/tmp/s360_fuzz', exist_ok=True)
import os, sys, json, subprocess
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
if len(sys.argv) > 2:
    from surround360_amd import _capi
    _capi.LIB_PATH = sys.argv[1]
    from surround360_amd import render as R, synth
    import numpy as np, rigutil
    import oracle_lib as O
    w,h,mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rig = R.RigDescription(ROOT + '/tests/golden/rig_17cam.json')
    rng = np.random.default_rng(w*100+h)
    i0 = rng.integers(0,256,(h,w,4),dtype=np.uint8); i1 = rng.integers(0,256,(h,w,4),dtype=np.uint8)
    i0[...,3] = 255; i1[...,3] = rng.choice([0,255,128],(h,w))
    if mode == 'oracle':
        want = O.compute_optical_flow(i0, i1, "pixflow_low", "LEFT"); print('ORACLE OK', want.shape); sys.exit(0)
    want = O.compute_optical_flow(i0, i1, "pixflow_low", "LEFT") if os.environ.get('NO_ORACLE') != '1' else None
    ctx = R.Context(rig, R.make_params(eqr_width=1008, eqr_height=504))
    ctx.set_sweep_mode(mode)
    try:
        got = ctx.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
        print('LIB OK' if want is None else ('EQUAL' if np.array_equal(got.view(np.uint32), want.view(np.uint32)) else 'DIFFER'))
    except _capi.S360Error as e:
        print('ERR', str(e)[:100])
    sys.exit(0)
res={}
for mode in sys.argv[2:] if False else os.environ.get('MODES','latency,throughput').split(','):
  for (w,h) in [(1,1),(2,2),(3,3),(4,4),(5,5),(6,6),(8,8),(2,40),(3,40),(4,40),(5,40),(40,2),(40,3),(40,4),(40,5),(7,9),(12,4),(4,12),(16,16),(1,40),(40,1),(49,49),(50,7)]:
    try:
        r = subprocess.run([sys.executable, __file__, sys.argv[1], str(w), str(h), mode], capture_output=True, text=True, timeout=int(os.environ.get('TMO','120')))
        out=(r.stdout.strip().splitlines() or ['<crash rc %d>' % r.returncode])[-1]
    except subprocess.TimeoutExpired:
        out='<HANG>' 
    print(mode, (w,h), out, flush=True)
