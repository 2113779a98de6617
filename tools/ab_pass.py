"""A/B of libraries on one box: ms per EM pass of a config for every EMSAR_HIP_LIB given (fresh process each, interleaved)."""
import os, subprocess, sys
libs = sys.argv[2:]
cfg = sys.argv[1]
code = ("import sys; sys.path.insert(0, %r)\n"
        "from emsar_amd import EmsarHip, synth\n"
        "s = synth.make_config(%r, 1.0)\n"
        "d = EmsarHip(0); d.upload_structure(s['n_tx'], s['row_ptr'], s['col_idx'], int(__import__('os').environ.get('AB_LAYOUT', '3'))); d.upload_sample(None, None, s['den']); d.run_passes(50)\n"
        "print(min(d.run_passes(200) / 200 for _ in range(5)))\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cfg)
for rep in range(2):
    for lib in libs:
        path, _, kv = lib.partition("@")               # library[@ENV=VALUE]
        env = dict(os.environ, EMSAR_HIP_LIB=os.path.abspath(path))
        for one in filter(None, kv.split(",")):           # library@A=1,B=2
            env[one.split("=")[0]] = one.split("=")[1]
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(lib, r.stdout.strip().splitlines()[-1], flush=True)
