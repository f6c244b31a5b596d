"""Run an UNCHANGED script of the reference (``main.py``) on the MI355X implementation:

    PYTHONPATH=/path/to/repo/doubly-contrastive-semseg_amd python -m dcs_amd.launch /path/to/reference/main.py [args...]

``python main.py`` puts the script's own directory first on ``sys.path``, ahead of ``PYTHONPATH``, so the reference's
``network`` / ``utils`` packages would win.  This launcher orders the path as [this package dir, the script's dir, ...]
and runs the script with ``runpy`` as ``__main__`` -- the merged drop-in packages (``_dropin.py``) then serve
``network.WeatherNet``, ``network.modeling.deeplabv3plus_*`` and ``utils.loss`` from here and everything else from the
reference tree.  No reference file is edited."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m dcs_amd.launch /path/to/reference/main.py [args...]")
    script = os.path.abspath(argv[0])
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_root = os.path.dirname(script)
    os.environ.setdefault("DCS_REFERENCE_ROOT", ref_root)
    rest = [p for p in sys.path if os.path.realpath(p or os.getcwd()) not in (os.path.realpath(here), os.path.realpath(ref_root))]
    sys.path[:] = [here, ref_root] + rest
    for name in ("network", "utils"):                      # a half-imported reference package must not linger
        for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
            f = getattr(sys.modules[k], "__file__", "") or ""
            if not os.path.realpath(f).startswith(os.path.realpath(here) + os.sep):
                del sys.modules[k]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
