"""Fold recordings of measured errors (SSLCR_RECORD_ERRORS=<file> python -m pytest tests -m gpu) into tests/measured_errors.json.

    python tools/update_measured.py [--merge | --reset] [--accept GLOB ...] [--accept-file FILE] [--report FILE] [--dry-run] REC.jsonl ...
    python tools/update_measured.py --diff-rev GIT_REV [--report FILE]        # the committed table against an older revision of itself

Per key the recordings' WORST value (several boxes / runs may be passed).  How it meets the current table:

  --merge   keys the recordings do not mention are kept; for the others the table takes max(old, recorded)  -- a ratchet, so:
  --reset   ... the table takes the RECORDED value, also where it is lower than the old one (a deliberate re-measurement after a
            numerics change: bounds may come DOWN as well as go up); keys the recordings do not mention are kept
  (neither) the table is replaced by the recordings

The table bounds the bf16 / fp8 assertions at 2 x its value, so a key that RISES loosens a test.  Every run prints the sorted diff
against the current table, and a key that rises by more than 1.5 x is REFUSED -- the table is not written and the exit status is 2 --
unless it is named by --accept (shell-style patterns; repeatable) or listed in --accept-file (one pattern per line, `#` starts a
comment: the place for the one-line cause that goes with each accepted rise; commit it as profiles/rNN_measured_diff.txt).  New keys
are listed and accepted."""
import fnmatch
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "measured_errors.json")
RISE = 1.5


def read_recordings(files):
    rec = {}
    for f in files:
        for line in open(f):
            line = line.strip()
            if not line:
                continue
            r = json.loads(line)
            rec[r["key"]] = max(rec.get(r["key"], 0.0), float(r["err"]))
    return rec


def diff_tables(old, new):
    """-> (rows, new_keys, gone_keys); rows = [(ratio, key, old, new)] for the keys whose value changed, largest rise first"""
    rows = []
    for k, v in new.items():
        if k in old and old[k] != v:
            rows.append((v / old[k] if old[k] > 0 else float("inf"), k, old[k], v))
    rows.sort(key=lambda r: (-r[0], r[1]))
    return rows, sorted(k for k in new if k not in old), sorted(k for k in old if k not in new)


def format_diff(rows, new_keys, gone_keys, accepted=()):
    out = []
    up = [r for r in rows if r[0] > 1.0]
    down = [r for r in rows if r[0] <= 1.0]
    out.append(f"{len(up)} keys rose ({sum(1 for r in up if r[0] > RISE)} by more than {RISE} x), {len(down)} fell, "
               f"{len(new_keys)} new, {len(gone_keys)} gone")
    for ratio, k, o, n in up:
        mark = "  " if ratio <= RISE else ("A " if any(fnmatch.fnmatch(k, p) for p in accepted) else "! ")
        out.append(f"{mark}{ratio:8.2f} x  {o:.4g} -> {n:.4g}  {k}")
    for ratio, k, o, n in down:
        out.append(f"  {ratio:8.2f} x  {o:.4g} -> {n:.4g}  {k}")
    for k in new_keys:
        out.append(f"+ new   {k}")
    for k in gone_keys:
        out.append(f"- gone  {k}")
    return "\n".join(out)


def take(argv, flag, many=False):
    vals = []
    while flag in argv:
        i = argv.index(flag)
        if i + 1 >= len(argv):
            raise SystemExit(f"{flag} needs a value")
        vals.append(argv[i + 1])
        del argv[i:i + 2]
    return vals if many else (vals[-1] if vals else None)


def main(argv):
    argv = list(argv)
    accept = take(argv, "--accept", many=True)
    accept_file = take(argv, "--accept-file")
    report = take(argv, "--report")
    rev = take(argv, "--diff-rev")
    if accept_file:
        for line in open(accept_file):
            line = line.split("#", 1)[0].strip()
            if line:
                accept.append(line.split()[0])
    cur = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if rev:                                      # report only: the committed table against an older revision of itself
        old = json.loads(subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:tests/measured_errors.json"], text=True))
        text = f"tests/measured_errors.json: {rev} -> working tree\n" + format_diff(*diff_tables(old, cur), accepted=accept)
        print(text)
        if report:
            open(report, "w").write(text + "\n")
        return 0
    merge, reset, dry = "--merge" in argv, "--reset" in argv, "--dry-run" in argv
    files = [a for a in argv if not a.startswith("--")]
    if not files:                                # (nothing to fold: never replace the table by an empty one)
        print(__doc__)
        return 0
    rec = read_recordings(files)
    if merge:
        table = dict(cur)
        for k, v in rec.items():
            table[k] = max(table.get(k, 0.0), v)
    elif reset:
        table = dict(cur)
        table.update(rec)
    else:
        table = rec
    table = {k: float(f"{v:.4g}") for k, v in sorted(table.items())}
    rows, new_keys, gone_keys = diff_tables(cur, table)
    text = format_diff(rows, new_keys, gone_keys, accepted=accept)
    print(text)
    if report:
        open(report, "w").write(text + "\n")
    refused = [r for r in rows if r[0] > RISE and not any(fnmatch.fnmatch(r[1], p) for p in accept)]
    if refused:
        print(f"\nREFUSED: {len(refused)} key(s) rise by more than {RISE} x and are not named by --accept / --accept-file "
              f"(marked `!` above); {OUT} is unchanged", file=sys.stderr)
        return 2
    if dry:
        print("(dry run: nothing written)")
        return 0
    with open(OUT, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
        f.write("\n")
    print(f"{OUT}: {len(table)} keys")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
