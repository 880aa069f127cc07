"""Phase timings of the small-map MBConv launches (mbmap.hip): one context, per launch, with the kernel's debug switches (BN_MM_DBG bit 1 =
no expand, 2 = no depthwise, 4 = no result stores) under both expand forms (BN_MBMAP_B3, or the switch named as the fourth argument).  Each configuration runs in a child process (the
switches are read at plan / launch time).  usage: python tools/mbmap_phases.py [v24|v30] [batch] [dbg,dbg,...] [switch]"""
import importlib, json, os, subprocess, sys, tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(model, batch):
    bn = importlib.import_module("rust-birdnet-onnx_amd")
    synth = importlib.import_module("rust-birdnet-onnx_amd.synth")
    p = tempfile.mktemp(suffix=".onnx")
    open(p, "wb").write({"v24": synth.birdnet_v24, "v30": synth.birdnet_v30}[model]())
    ctx = bn.Context(bn.Model(p), batch)
    ctx.time_kernels(batch)
    runs = [ctx.time_kernels(batch) for _ in range(3)]
    out = [(runs[0][i][0], min(r[i][1] for r in runs)) for i in range(len(runs[0])) if runs[0][i][0].startswith("mbconv:")]
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    model = sys.argv[1] if len(sys.argv) > 1 else "v24"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    DBGS = (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1", "2", "3", "4"])
    SWITCH = sys.argv[4] if len(sys.argv) > 4 else "BN_MBMAP_B3"
    table = {}
    for b3 in ("1", "0"):
        for dbg in DBGS:
            env = dict(os.environ, BN_MM_DBG=dbg)
            env[SWITCH] = b3
            r = subprocess.run([sys.executable, __file__, "--child", model, str(batch)], env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print(r.stderr[-2000:]); sys.exit(1)
            table[(b3, dbg)] = json.loads(r.stdout.strip().splitlines()[-1])
    names = [n for n, _ in table[("1", DBGS[0])]]
    print(f"{'launch':34s} " + f"{SWITCH}: " + " ".join(f"{b}/dbg={d}" for b in "10" for d in DBGS))
    for i, n in enumerate(names):
        print(f"{n[:34]:34s} " + " ".join(f"{table[(b, d)][i][1]:11.1f}" for b in "10" for d in DBGS))
