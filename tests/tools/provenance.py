"""Which tree a profile summary was measured on: the commit, the date and a hash of pywindow_amd/csrc/* go INTO the
summaries under profiles/ (JSON: a "provenance" key; CSV: a .provenance.json sidecar), so that bench.py -- which
quotes those summaries as static inputs of its line -- can say where they come from and flag them when the kernels
have changed since ("stale": the hash of the csrc files differs).

    python tests/tools/provenance.py stamp                  # here, before gpurun: .pw_head (the GPU box has no .git)
    python tests/tools/provenance.py annotate FILE [...]    # on the GPU box, at the end of a profile round
    python tests/tools/provenance.py show                   # the current tree's stamp
"""
import datetime
import hashlib
import json
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]


def csrc_sha16(root=ROOT) -> str:
    h = hashlib.sha256()
    for f in sorted((pathlib.Path(root) / "pywindow_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".hpp", ".cpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def head_info(root=ROOT) -> dict:
    root = pathlib.Path(root)
    try:
        head = subprocess.run(["git", "-C", str(root), "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", str(root), "status", "--porcelain", "--", "pywindow_amd", "bench.py"],
                                    capture_output=True, text=True, check=True).stdout.strip())
        return {"head": head[:12], "dirty": dirty}
    except Exception:
        try:
            return json.loads((root / ".pw_head").read_text())
        except Exception:
            return {"head": None, "dirty": None}


def stamp(root=ROOT) -> dict:
    info = head_info(root)
    info.update({"csrc_sha16": csrc_sha16(root), "date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ")})
    return info


def main(argv):
    if len(argv) < 2 or argv[1] == "show":
        print(json.dumps(stamp()))
    elif argv[1] == "stamp":
        info = head_info()
        (ROOT / ".pw_head").write_text(json.dumps(info))
        print(json.dumps(info))
    elif argv[1] == "annotate":
        info = stamp()
        for name in argv[2:]:
            p = pathlib.Path(name)
            if not p.exists():
                continue
            if p.suffix == ".json":
                try:
                    d = json.loads(p.read_text())
                except ValueError:
                    continue
                d["provenance"] = info
                p.write_text(json.dumps(d, indent=1))
            else:
                pathlib.Path(str(p) + ".provenance.json").write_text(json.dumps(info))
        print(json.dumps(info))


if __name__ == "__main__":
    main(sys.argv)
