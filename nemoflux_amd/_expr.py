"""Safe evaluation of the small expression strings nemoflux takes on its command lines.

The reference passes them to eval() (datagen.py:78, fluxexact.py:41-43, fluxplot.py:30, fluxviz.py:378).  Here point
lists are parsed as pure literals, and stream functions are checked node by node: arithmetic on the names x, y, z, t,
nt, pi and calls of a fixed set of numpy functions -- nothing else compiles."""
import ast

import numpy

FUNCTIONS = {'cos': numpy.cos, 'sin': numpy.sin, 'tan': numpy.tan, 'arctan2': numpy.arctan2, 'arctan': numpy.arctan,
             'exp': numpy.exp, 'log': numpy.log, 'sqrt': numpy.sqrt, 'abs': numpy.abs, 'tanh': numpy.tanh,
             'cosh': numpy.cosh, 'sinh': numpy.sinh}
CONSTANTS = {'pi': numpy.pi}
_NODES = (ast.Expression, ast.BinOp, ast.UnaryOp, ast.Call, ast.Name, ast.Constant, ast.Load, ast.Add, ast.Sub, ast.Mult,
          ast.Div, ast.Pow, ast.Mod, ast.USub, ast.UAdd)


def literal(text, what='value'):
    """A Python literal (numbers, tuples, lists, strings): the -l / -i / --deltaDeg arguments."""
    try:
        return ast.literal_eval(text.strip())
    except (ValueError, SyntaxError, MemoryError, RecursionError) as e:
        raise RuntimeError(f'ERROR: cannot parse {what} {text!r}: {e}') from e


def compile_function(text, variables=('x', 'y', 'z', 't', 'nt')):
    """Code object of an arithmetic expression in `variables`; RuntimeError for anything else."""
    try:
        tree = ast.parse(text.strip(), mode='eval')
    except SyntaxError as e:
        raise RuntimeError(f'ERROR: cannot parse expression {text!r}: {e}') from e
    for node in ast.walk(tree):
        if not isinstance(node, _NODES):
            raise RuntimeError(f'ERROR: {type(node).__name__} is not allowed in expression {text!r}')
        if isinstance(node, ast.Call):
            if not isinstance(node.func, ast.Name) or node.func.id not in FUNCTIONS or node.keywords:
                raise RuntimeError(f'ERROR: only calls of {sorted(FUNCTIONS)} are allowed in expression {text!r}')
        elif isinstance(node, ast.Name):
            if node.id not in FUNCTIONS and node.id not in CONSTANTS and node.id not in variables:
                raise RuntimeError(f'ERROR: unknown name {node.id!r} in expression {text!r}')
        elif isinstance(node, ast.Constant):
            if not isinstance(node.value, (int, float)) or isinstance(node.value, bool):
                raise RuntimeError(f'ERROR: only numeric constants are allowed in expression {text!r}')
    return compile(tree, '<expression>', 'eval')


def evaluate(code, **values):
    env = dict(FUNCTIONS)
    env.update(CONSTANTS)
    env.update(values)
    return eval(code, {'__builtins__': {}}, env)   # `code` passed compile_function's node check
