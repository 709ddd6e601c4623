"""What a counter summary under profiles/ was measured on: the hash of the library sources in THIS tree (the tree the
session ran from) and the A/B switches in the environment.  bench.py attaches a summary's numbers only when the
stamp's `src` equals the source hash compiled into the library it has loaded."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def stamp():
  from graphcast_amd import _native
  return {"src": _native.source_hash(),
          "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GCAST_")}}
