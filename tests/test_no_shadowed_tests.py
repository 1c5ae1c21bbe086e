"""A second `def test_x` in a module silently REPLACES the first one: pytest only ever sees the last binding, and the first test's
cases never run (round 3: tests/test_gpu_vae.py carried two `test_downsample_conv_vs_torch`).  This check parses every test module and
fails on duplicate function / class names at module level and duplicate method names inside a test class."""
import ast
import glob
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _duplicates(body):
    seen, dup = {}, []
    for node in body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            if node.name in seen:
                dup.append((node.name, seen[node.name], node.lineno))
            seen[node.name] = node.lineno
    return dup


def test_no_test_module_defines_a_name_twice():
    problems = []
    for path in sorted(glob.glob(os.path.join(HERE, "*.py"))):
        tree = ast.parse(open(path).read(), filename=path)
        for name, first, second in _duplicates(tree.body):
            problems.append(f"{os.path.basename(path)}: `{name}` defined at line {first} and again at line {second}")
        for node in tree.body:
            if isinstance(node, ast.ClassDef):
                for name, first, second in _duplicates(node.body):
                    problems.append(f"{os.path.basename(path)}: `{node.name}.{name}` defined at line {first} and again at line {second}")
    assert not problems, "shadowed definitions (the earlier one never runs):\n" + "\n".join(problems)
