import sys
sys.path.insert(0, '/root/repo/nubomedia-vca_amd')
from nubovca import capi, synth
ctx = capi.Context(0); casc = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
g = synth.make_gray(150, 100, 1, "natural")
ctx.detect_multiscale(casc, g, 1.1, 2, capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1))
print("---- second call", file=sys.stderr)
ctx.detect_multiscale(casc, g, 1.1, 2, capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1))
