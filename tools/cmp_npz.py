import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    d = np.abs(a[k] - b[k])
    print("%-10s max|diff| %.3e  at %s  shape %s" % (k, d.max(), np.unravel_index(d.argmax(), d.shape), a[k].shape))
