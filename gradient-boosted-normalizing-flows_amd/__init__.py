"""MI355X-native boosted normalizing-flow density evaluator (hot path only).

Import name: ``gbnf_amd`` (the on-disk directory keeps the upstream project's
hyphenated name, which Python cannot import directly; ``gbnf_amd/__init__.py``
at the repo root aliases it).

Public surface (mirrors models/boosted_flow.py of the reference):
  BoostedFlow(args)                         -- drop-in host module, forward(x=, components=)
  component_forward / component_log_prob / log_prob  -- convenience names of BASELINE.json
  native                                    -- ctypes binding of libgbnf_hip.so (include/gbnf.h)
  checkpoint.save / checkpoint.load         -- the reference's checkpoint format (utils/utilities.py:42-93) + side-car
"""
from . import spec, synth  # noqa: F401  (pure-python, always importable)

__all__ = ["spec", "synth", "native", "BoostedFlow"]


def __getattr__(name):
    # lazy so that ``import gbnf_amd`` works on a box where torch / the .so are absent;
    # anything that computes fails loudly inside ``native``.
    import importlib
    if name in ("native", "boosted_flow", "sharded", "checkpoint", "image_glow"):
        return importlib.import_module(__name__ + "." + name)
    if name == "BoostedFlow":
        return importlib.import_module(__name__ + ".boosted_flow").BoostedFlow
    raise AttributeError(name)
