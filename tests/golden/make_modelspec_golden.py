"""Writes tests/golden/modelspec_<name>.h: the generated model plugin of the five shipped example models, as text.
The files are committed after being read against the templates of compile.m:255-551; tests/test_plugin_independent.py
compares the generator's output with them.  Data for the tests only: nothing is compiled from these copies."""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from egdst_amd import codegen, examples  # noqa: E402

if __name__ == '__main__':
    for name in ('deaton1', 'deaton2', 'retirement1', 'retirement2', 'occ3'):
        text = codegen.generate_modelspec(examples.REGISTRY[name]())
        open(os.path.join(HERE, 'modelspec_%s.h' % name), 'w').write(text)
        print(name, len(text.splitlines()), 'lines')
