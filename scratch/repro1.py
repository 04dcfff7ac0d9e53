import sys, os, numpy as np, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imcoalhmm_amd import Forwarder, _capi, synth
d = np.load('tests/golden/hmm_params.npz')
pi, T, E = d['iso20_t0_pi'], d['iso20_t0_T'], d['iso20_t0_E']
L = _capi.lib()
obs = synth.sample_alignment(pi, T, E, 10_000_000, seed=1)
f = Forwarder.from_array(obs, 3)
print('a', f.forward(pi, T, E), _capi.last_plan(), flush=True)
L.imc_set_segment_length(50_000)
print('b', f.forward(pi, T, E), _capi.last_plan(), flush=True)
L.imc_set_segment_length(0)
L.imc_set_compression(0)
print('c', Forwarder.from_array(obs, 3).forward(pi, T, E), _capi.last_plan(), flush=True)
L.imc_set_compression(1)
head = obs[:2_000_000]
g = Forwarder.from_array(head, 3)
print('created', g.compressed_length(), flush=True)
print('d', g.forward(pi, T, E), _capi.last_plan(), flush=True)
