import sys, os, cProfile, pstats, io
sys.argv = ['config5_probe.py', '43', '6']
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
pr = cProfile.Profile(); pr.enable()
exec(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools', 'config5_probe.py')).read())
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
