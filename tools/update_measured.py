"""Fold a recording of measured errors (SSLCR_RECORD_ERRORS=<file> python -m pytest tests -m gpu) into tests/measured_errors.json:
per key the WORST value over all recordings given (several boxes / runs may be passed).  `--merge` keeps keys of the current
table that the recordings do not mention and takes the max for the others; without it the table is replaced."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "measured_errors.json")


def main(argv):
    merge = "--merge" in argv
    files = [a for a in argv if not a.startswith("--")]
    if not files:                                # (nothing to fold: never replace the table by an empty one)
        print(__doc__)
        return
    table = {}
    if merge and os.path.exists(OUT):
        table = json.load(open(OUT))
    for f in files:
        for line in open(f):
            line = line.strip()
            if not line:
                continue
            r = json.loads(line)
            table[r["key"]] = max(table.get(r["key"], 0.0), float(r["err"]))
    with open(OUT, "w") as f:
        json.dump({k: float(f"{v:.4g}") for k, v in sorted(table.items())}, f, indent=0, sort_keys=True)
        f.write("\n")
    print(f"{OUT}: {len(table)} keys")


if __name__ == "__main__":
    main(sys.argv[1:])
