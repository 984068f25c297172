"""Loads the hot-path functions of the reference scripts for golden-vector generation.

TEST INFRASTRUCTURE ONLY, and only usable in the build container where the
reference checkout is mounted at /root/reference.  Nothing here is shipped
or imported by tests that run on the GPU box: the vectors it produces are
committed under tests/golden/ together with make_golden.py.

The reference scripts cannot be imported as modules (top-level code reads data
files and imports a package that is not installed), so the file is parsed and
only its function definitions, plain imports and the scalar flag assignments
the walk reads as globals are executed.  No reference source text is copied.
"""
import ast
import os
import sys
import types

REFERENCE = "/root/reference"

_MC_FLAGS = {"x_init", "y_init", "z_init", "phi_init", "mu_init", "lambda_init", "g_av",
             "albedo", "Q_ext", "tau_scale", "initial_state", "rand_mu", "abs_only",
             "monochromatic", "alpha", "h", "wl_min", "wl_max", "min_tau", "max_tau"}
_BANNED_IMPORTS = ("hoki", "read_BPASS", "matplotlib")


def _keep(node, flags):
    if isinstance(node, ast.FunctionDef):
        return True
    if isinstance(node, (ast.Import, ast.ImportFrom)):
        text = ast.dump(node)
        return not any(b in text for b in _BANNED_IMPORTS)
    if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
        return node.targets[0].id in flags
    return False


def load(filename, flags=_MC_FLAGS, name=None):
    path = os.path.join(REFERENCE, filename)
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if _keep(n, flags)]
    mod = types.ModuleType(name or ("ref_" + os.path.splitext(filename)[0].replace(" ", "_")))
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        pass
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), mod.__dict__)
    return mod


def load_mc_code():
    return load("MC code.py")


def load_gray():
    return load("Numpy_random_walk.py")


def load_read_bpass():
    return load("read_BPASS.py", flags=set())
