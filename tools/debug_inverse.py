"""Debug aid: per-prefix inverse error on the device vs the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import GoldenCase
from gbnf_amd import native
from oracle import gbnf_oracle as oracle

for name in sys.argv[1:]:
    g = GoldenCase(name)
    spec = g.specs[0]
    for K in range(1, len(spec["steps"]) + 1):
        sp = dict(spec); sp["steps"] = spec["steps"][:K]
        z64, ldj64 = oracle.component_forward(sp, g.x, backend="numpy64")
        z = np.ascontiguousarray(z64, dtype=np.float32)
        x_or, ild_or = oracle.component_inverse(sp, z, backend="numpy64")
        for math in ("f32",):
            f = native.NativeFlow(sp, math=math)
            x, ild = f.inverse(torch.from_numpy(z).cuda())
            e = np.abs(x.cpu().numpy() - x_or)
            print(name, "K", K, "oracle rt", np.abs(x_or - g.x).max(), "dev err", e.max(), "worst feature", e.max(0).argmax(),
                  "ild err", np.abs(ild.cpu().numpy() - ild_or).max(), "per-feature", np.round(e.max(0), 6)[:24])
    if os.environ.get("DBG_DETAIL"):
        sp = dict(spec); sp["steps"] = spec["steps"][:1]
        z64, _ = oracle.component_forward(sp, g.x, backend="numpy64")
        z = np.ascontiguousarray(z64, dtype=np.float32)
        x_or, _ = oracle.component_inverse(sp, z, backend="numpy64")
        x, _ = native.NativeFlow(sp, math="f32").inverse(torch.from_numpy(z).cuda())
        x = x.cpu().numpy()
        for f in (12, 13, 17):
            print("feature", f, "z", z[:6, f], "\n   x_dev", x[:6, f], "\n   x_or ", x_or[:6, f])
