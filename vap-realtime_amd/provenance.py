"""Which kernel sources a measurement was taken with.  The GPU box has no ``.git`` (gpurun ships a snapshot), so the identity that travels is a
content hash of the library's sources; the git commit is recorded next to it where a repository is present."""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def _hash_files(named_blobs) -> str:
    h = hashlib.sha256()
    for name, blob in sorted(named_blobs):
        h.update(name.encode() + b"\0" + hashlib.sha256(blob).digest())
    return h.hexdigest()[:16]


def kernel_source_hash(csrc: str = CSRC) -> str:
    """sha256 (16 hex digits) over every ``*.hip / *.h / *.cpp`` of ``csrc`` (names + contents; generated vapx_layout.h excluded)."""
    files = [p for pat in ("*.hip", "*.h", "*.cpp") for p in glob.glob(os.path.join(csrc, pat)) if os.path.basename(p) != "vapx_layout.h"]
    return _hash_files((os.path.basename(p), open(p, "rb").read()) for p in files)


def kernel_source_hash_at(commit: str, repo: str) -> Optional[str]:
    """The same hash for the tree of a git commit (stamping PMC passes taken at an earlier tree)."""
    try:
        names = subprocess.run(["git", "-C", repo, "ls-tree", "--name-only", commit, "vap-realtime_amd/csrc/"], check=True, capture_output=True,
                               text=True).stdout.split()
        blobs = []
        for n in names:
            b = os.path.basename(n)
            if b.endswith((".hip", ".h", ".cpp")) and b != "vapx_layout.h":
                blobs.append((b, subprocess.run(["git", "-C", repo, "show", f"{commit}:{n}"], check=True, capture_output=True).stdout))
        return _hash_files(blobs)
    except Exception:      # noqa: BLE001
        return None


def git_head(repo: str) -> Optional[str]:
    try:
        return subprocess.run(["git", "-C", repo, "rev-parse", "--short=12", "HEAD"], check=True, capture_output=True, text=True).stdout.strip() or None
    except Exception:      # noqa: BLE001
        return None
