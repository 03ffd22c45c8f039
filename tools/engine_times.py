import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t=time.time()
from clair3_rna_amd import capi, synth
print("import %.3f" % (time.time()-t))
w = synth.random_weights(18)
for rep in range(3):
    t0=time.time(); e=capi.Engine(0); t1=time.time(); e.load_weights(w,18); t2=time.time(); e.set_precision("f16x3"); t3=time.time()
    print("Engine %.3f s, load_weights %.3f s, set_precision %.3f s" % (t1-t0, t2-t1, t3-t2))
    e.close()
