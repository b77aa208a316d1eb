"""BASELINE.json configs[4]: a 1 GiB mixed corpus (4 x 256 MiB: text, random, DNA, 1000-byte motif repeated) through
the CLI's multi-block container: `archon e -b256m`, `archon d -b`, byte compare; plus the device-side times of each
block's forward and inverse through the C ABI; then the same corpus through `archon e -m` (the config's "MTF/entropy
stage": no reference implementation, SURVEY.md 8(f) N4 -- on the GPU behind the transform) and back.
Usage: python tools/config5.py [block MiB]"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np
import archon_synth as S, pyarchon

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
shapes = ["text", "random", "dna", "motif"]
tmp = "/tmp/config5_%d" % os.getpid()
res = {"block_mib": mib, "blocks": []}
with open(tmp + ".in", "wb") as f:
    for b, sh in enumerate(shapes):
        x = S.gen_shape(sh, n, block=b)
        x.tofile(f)
        sa, bwt, base = pyarchon.forward(x)            # first call of a route: includes the one-time load of its kernels
        cold_ms = pyarchon.stats()["ms_total"]
        del sa, bwt
        sa, bwt, base = pyarchon.forward(x)
        st = pyarchon.stats()
        back = pyarchon.inverse(bwt, base)
        si = pyarchon.stats()
        assert (back == x).all() and pyarchon.validate(x, sa)
        res["blocks"].append({"shape": sh, "forward_ms": round(st["ms_total"], 3), "forward_first_call_ms": round(cold_ms, 3), "path": st["path"],
                              "doubling_rounds": st["doubling_rounds"], "inverse_ms": round(si["ms_total"], 3)})
        del sa, bwt, back, x
exe = os.path.join(ROOT, "bin", "archon")
t0 = time.time(); r = subprocess.run([exe, "e", "-b%dm" % mib, tmp + ".in", tmp + ".ra"], capture_output=True, text=True); t1 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
r = subprocess.run([exe, "d", "-b", tmp + ".ra", tmp + ".out"], capture_output=True, text=True); t2 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
def sha(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for c in iter(lambda: f.read(1 << 24), b""):
            h.update(c)
    return h.hexdigest()
res["cli_round_trip_identical"] = sha(tmp + ".in") == sha(tmp + ".out")
res["cli_encode_wall_s"] = round(t1 - t0, 2); res["cli_decode_wall_s"] = round(t2 - t1, 2)
res["container_bytes"] = os.path.getsize(tmp + ".ra")
# the same corpus with the MTF + zero-run + Huffman stage (`-m`; on the GPU behind the transform, csrc/post.hiph; decoded by host threads)
t0 = time.time(); r = subprocess.run([exe, "e", "-m", "-b%dm" % mib, tmp + ".in", tmp + ".rm"], capture_output=True, text=True); t1 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
r = subprocess.run([exe, "d", "-b", tmp + ".rm", tmp + ".out2"], capture_output=True, text=True); t2 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
res["post_round_trip_identical"] = sha(tmp + ".in") == sha(tmp + ".out2")
res["post_encode_wall_s"] = round(t1 - t0, 2); res["post_decode_wall_s"] = round(t2 - t1, 2)
res["post_container_bytes"] = os.path.getsize(tmp + ".rm")
os.remove(tmp + ".rm"); os.remove(tmp + ".out2")
res["device_forward_ms_total"] = round(sum(b["forward_ms"] for b in res["blocks"]), 2)
res["device_inverse_ms_total"] = round(sum(b["inverse_ms"] for b in res["blocks"]), 2)
for ext in (".in", ".ra", ".out"):
    os.remove(tmp + ext)
print(json.dumps(res))
assert res["cli_round_trip_identical"] and res["post_round_trip_identical"]
