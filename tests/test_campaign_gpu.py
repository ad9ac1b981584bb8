"""The random-seed parity campaigns of tools/ (the runs DESIGN.md section 6 quotes take minutes to hours), bounded to about half a
minute each and pinned to fixed first seeds, so that a plain `pytest -m gpu` run on a fresh box carries a few thousand streams and a
few hundred thousand chunks of its own evidence: every GPU output byte for byte against the oracle / the reference built in place,
every malformed stream accepted or refused as the checker decides.  TEST INFRASTRUCTURE (the tools load oracle/)."""
import os, re, subprocess, sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CAMPAIGNS = [
    # tool, arguments (seconds, first seed, ...), what the last line must say, the count it must report at least
    ("fuzz_campaign.py", ["30", "7101"], r"fuzz campaign OK: (\d+) inputs", 5),                       # all codecs, table modes, Zstd levels 1 - 4
    ("fuzz_entropy_campaign.py", ["25", "7102"], r"OK seed0=7102 batches=(\d+)", 1),                   # Zstd entropy stage, prescribed histograms
    ("fuzz_decode_campaign.py", ["30", "7103"], r"decode fuzz campaign OK: (\d+) streams", 256),       # mutated streams of the three codecs
    ("fuzz_decode_frames_campaign.py", ["30", "7104"], r"multi-block decode fuzz campaign OK: (\d+) streams", 256),
    ("fuzz_foreign_decode_campaign.py", ["25", "7105"], r"OK seed0=7105 inputs=(\d+)", 10),            # streams the reference wrote: levels -7 .. 22, LZ4HC
]


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args,pattern,at_least", CAMPAIGNS, ids=[c[0][:-3] for c in CAMPAIGNS])
def test_bounded_campaign(tool, args, pattern, at_least):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    m = re.search(pattern, r.stdout)
    assert m, tail
    assert int(m.group(1)) >= at_least, tail
