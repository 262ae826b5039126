"""Which form should an irregular matrix get when reruns must be bit-identical (the exact fold as the default, the relaxed order as
an opt-in)?  Per family of the zoo: CU-wide tiles in relaxed order (round 5's default) | CU-wide tiles with rows owned by wavefronts (exact, round 6) |
wavefront-private exact tiles (round 4) | no tiles (lflat / column panels / plain CSR: whatever the ladder picks then)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import form_tournament as ft
FORMS = (("default", {}), ("relaxed CU tiles", {"tile_relaxed": 1}), ("row-owned exact CU tiles", {"tile_relaxed": 0}),
         ("wavefront-private exact tiles", {"tile_relaxed": 0, "tile_rowown": 0}), ("no tiles", {"tiles": 0}), ("no tiles, no lflat", {"tiles": 0, "lflat": 0}))
for name, scale in (("random_spd_1m", None), ("rand100", 0.5), ("rand200", 1.0), ("rand500", 0.6), ("powerlaw", 0.5), ("dense_rows_200k", None), ("random_spd_10m", None)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    ft.run(name, 30, FORMS, out=lambda t: print(t, flush=True), scale=scale)
