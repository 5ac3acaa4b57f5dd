"""Random rank triples / iteration counts / bounds on a 136-image batch: the round-5 paths forced on (LRF_PERSIST=1: k_bcd_p with
the first iteration inside at ranks <= 16; LRF_FUSED_GRAM_MIN_CHUNKS=1: k_planes16_gram) against both off (LRF_PERSIST=0, the
two-kernel front end): digests of the int8 factors must agree case by case.  Uses dev_fuzz_families.py's child mode.
python tools/dev_fuzz_round5.py [cases] [seed]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
cases = sys.argv[1] if len(sys.argv) > 1 else "30"
seed = sys.argv[2] if len(sys.argv) > 2 else "7"
outs = []
for extra in ({"LRF_PERSIST": "1", "LRF_FUSED_GRAM_MIN_CHUNKS": "1"}, {"LRF_PERSIST": "0", "LRF_FUSED_GRAM_MIN_CHUNKS": str(1 << 40)}):
    r = subprocess.run([sys.executable, os.path.join(HERE, "dev_fuzz_families.py"), "child", cases, seed], env=dict(os.environ, **extra),
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:])
        sys.exit(2)
    outs.append([l for l in r.stdout.splitlines() if l.startswith("(")])
bad = [(a, b) for a, b in zip(*outs) if a != b]
for a, b in bad:
    print("MISMATCH", a, "|", b)
print(f"round-5 fuzz: {len(outs[0])} cases (seed {seed}), {len(bad)} mismatches")
sys.exit(1 if bad or len(outs[0]) != int(cases) or len(outs[1]) != int(cases) else 0)
