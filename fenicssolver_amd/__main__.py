"""``python -m fenicssolver_amd case.json`` (reference: FenicsSolver/__init__.py:12-13,
main.py:97-107)."""
import sys
from .main import main

if __name__ == "__main__":
    if len(sys.argv) < 2:
        print("Not enough input argument, Usage: `python -m fenicssolver_amd case_input.json`")
        sys.exit(2)
    main(sys.argv[1])
