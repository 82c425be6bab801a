"""Hash of everything that determines what the kernels do (csrc/*.hip, common.h, include/yolohip.h).  Profiles under
profiles/ are stamped with it; bench.py only quotes a stored measurement (HBM traffic) when the stamp matches the tree it
runs from, so a number can never silently outlive the kernels it was taken on."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash() -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "yolo-from-scratch_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "yolo-from-scratch_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "yolohip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def git_head() -> str:
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    except OSError:
        return "unknown"


def stamp() -> dict:
    return {"csrc_sha16": csrc_hash(), "git_head": git_head()}


if __name__ == "__main__":
    print(stamp())
