import sys
sys.path.insert(0, "/root/repo")
from ihm2_amd import _lib
import os
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
