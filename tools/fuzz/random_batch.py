"""s360_frame_render_batch with random flags: 2..4 different frames in the slots of one context (their flows in the same
batched kernels, the throughput sweep kernel), two chained batches (temporal state per slot) and — round 5 — a third chained step for
a random SUBSET of the slots (s360_frame_render_slots: streams of unequal length), on an emulated build of the library; every slot's
stereo equirect against the oracle rendering that frame chain alone.
usage: python tools/fuzz/random_batch.py <libs360 build> <seed> <cases>"""
import json, os, subprocess, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
if len(sys.argv) > 2 and sys.argv[2] == 'one':
    from surround360_amd import _capi
    _capi.LIB_PATH = sys.argv[1]
    from surround360_amd import render as R
    import numpy as np, rigutil
    import oracle_lib as O
    kw = json.loads(sys.argv[3]); cam = kw.pop('cam'); slots = kw.pop('slots'); subset = kw.pop('subset')
    os.makedirs('/tmp/s360_fuzz', exist_ok=True)
    path = rigutil.scaled_rig_json(ROOT + '/tests/golden/rig_17cam.json', '/tmp/s360_fuzz/rig_batch_%d.json' % cam, cam / 2048.0)
    rig = R.RigDescription(path)
    cams, _ = O.load_rig(path)
    okw = {k: v for k, v in kw.items() if not k.endswith('_alg')}
    okw['side_flow_search20'] = int(kw['side_flow_alg'] == 'pixflow_search_20')
    frames = [[rigutil.frame_inputs(path, cam, yaw_deg=1.3 * s + 0.4 * f, world_h=256) for f in range(3)] for s in range(slots)]
    ctx = R.Context(rig, R.make_params(**kw))
    ctx.set_frame_slots(slots); ctx.set_sweep_mode("throughput")
    got = [[None, None, None] for _ in range(slots)]
    for f in range(3):
        live = list(range(slots)) if f < 2 else subset
        for s in live:
            ctx.select_frame_slot(s); ctx.upload_frame(*frames[s][f])
        if f < 2: ctx.render_batch(use_prev=(f > 0))
        else: ctx.render_slots(live, use_prev=True)
        for s in live:
            ctx.select_frame_slot(s); got[s][f] = ctx.download_equirect()
    ok = True
    for s in range(slots):
        of = O.Frame(cams, O.make_params(**okw))
        for f in range(3 if s in subset else 2):
            want, _ = of.render(*frames[s][f], use_prev=(f > 0))
            ok = ok and got[s][f].shape == want.shape and bool(np.array_equal(got[s][f], want))
    print('EQUAL' if ok else 'DIFFER')
    sys.exit(0)
random.seed(int(sys.argv[2])); n = int(sys.argv[3]); bad = 0
for i in range(n):
    cam = random.choice([48, 64, 96])
    w = 14 * random.randint(3, 24); h = random.randint(70, 170)  # (pole images taller than the feather: the reference's domain)
    poles = random.choice([0, 1, 1])
    kw = dict(cam=cam, slots=random.randint(2, 4), eqr_width=w, eqr_height=h, enable_top=poles, enable_bottom=poles,
              final_eqr_width=random.choice([0, 0, 120]), final_eqr_height=random.choice([0, 0, 96]),
              sharpening=random.choice([0.0, 0.25]), side_alpha_feather_size=random.randint(0, cam // 2),
              std_alpha_feather_size=random.choice([3, 9, 15, 31]), interpupilary_dist=random.choice([6.4, 3.0]),
              zero_parallax_dist=random.choice([10000.0, 300.0]), side_flow_alg=random.choice(["pixflow_low", "pixflow_search_20"]),
              polar_flow_alg="pixflow_low")
    if (kw['final_eqr_width'] == 0) != (kw['final_eqr_height'] == 0): kw['final_eqr_width'] = kw['final_eqr_height'] = 0
    kw['subset'] = sorted(random.sample(range(kw['slots']), random.randint(1, kw['slots'] - 1)))
    try:
        r = subprocess.run([sys.executable, __file__, sys.argv[1], 'one', json.dumps(kw)], capture_output=True, text=True, timeout=1500)
        outl = [l for l in r.stdout.strip().splitlines() if l]
        status = outl[-1] if outl else 'rc %d %s' % (r.returncode, (r.stderr.strip().splitlines() or [''])[-1][:160])
    except subprocess.TimeoutExpired:
        status = 'TIMEOUT'
    if status != 'EQUAL':
        bad += 1
        print(i, status[:160], json.dumps(kw), flush=True)
print('done: %d cases, %d not equal' % (n, bad))
