"""profiles/<tag>_parity_errors.txt from a `pytest -m gpu` log: the measured-error table tests/conftest.py prints (plus the recorded device times and the outcome line)."""
import re, subprocess, sys
log, tag = sys.argv[1], sys.argv[2]
lines = open(log).read().splitlines()
start = next(i for i, l in enumerate(lines) if "measured parity errors" in l)
outcome = [l for l in lines if re.search(r"\d+ passed", l)][-1]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
body = [l for l in lines[start:] if l.strip() and not l.startswith("rc=")]
with open("profiles/%s_parity_errors.txt" % tag, "w") as f:
    f.write("# python -m pytest tests/ -x -q -m gpu on MI355X (round %s, HEAD %s): %s\n" % (tag.lstrip("r0"), head, outcome.strip("= ")))
    f.write("# columns: test, tensor, max elementwise error |a-b| / (|b| + floor) against its tolerance, the PLAIN max|a-b|/max|b| of the tensor, cases, elements, excluded (fragile) pixels / rays;\n")
    f.write("# gradient rows also carry the same comparison at the other multiples of the oracle's measured fp32 uncertainty (K_UNC; 0 = none)\n")
    f.write("\n".join(body) + "\n")
print("wrote profiles/%s_parity_errors.txt (%d lines)" % (tag, len(body) + 3))
