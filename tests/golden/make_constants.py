"""Generates tests/golden/constants.json by IMPORTING the reference's python/constants.py
(numpy-only, importable in the build container) and dumping its public floats.
Run in the build container only: /root/reference does not exist on the GPU box.

    python tests/golden/make_constants.py
"""
import importlib.util
import json
import os

REF = "/root/reference/python/constants.py"
spec = importlib.util.spec_from_file_location("ref_constants", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
out = {}
for name in sorted(dir(mod)):
    if name.startswith("_"):
        continue
    v = getattr(mod, name)
    if isinstance(v, (int, float)) or type(v).__name__ in ("float64",):
        out[name] = float(v)
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "constants.json"), "w") as f:
    json.dump({"source": "python/constants.py (reference @ v2)", "values": out}, f, indent=1, sort_keys=True)
print(len(out), "constants written")
