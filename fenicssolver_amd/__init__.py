"""fenicssolver_amd — MI355X-native assemble + Krylov-solve pipeline behind the
FenicsSolver Python API (SolverBase / ScalarTransportSolver /
LinearElasticitySolver, JSON case settings).

Mirrors FenicsSolver/__init__.py:9-13 of the reference, except that importing
the package never starts a solve by itself (the reference runs ``main(sys.argv)``
on import when argv has >= 2 entries — SURVEY.md Appendix B-Q1); use
``python -m fenicssolver_amd case.json`` instead.
"""
__version__ = "0.1"

from .main import main, load_settings  # noqa: F401
