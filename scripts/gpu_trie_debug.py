import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from painlessinferenceacceleration_amd.device_trie import DeviceTrie
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from tests import trie_replay as tr
path = [p for p in tr.trace_files() if 'trace_2' in p][0]
trace = tr.load(path); init = trace['init']
cache = LookaheadCache(eos_ids=init['eos_ids'], stop_words={w: 1 for w in init['stop_words']}, max_node=init['max_node'], max_output_node=init['max_output_node'])
for i, op in enumerate(trace['ops']):
    name = op['op']
    if name == 'put': cache.put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], mode=op['mode'], idx=op['idx'])
    elif name == 'stream_put': cache.stream_put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], idx=op['idx'])
    elif name == 'reset_input_freqs': cache.reset_input_freqs(op['idx'])
    elif name == 'squeeze_branch_counts': cache.squeeze_branch_counts()
    elif name == 'fresh': cache.fresh()
    elif name == 'limits': cache.max_node, cache.max_output_node = op['max_node'], op['max_output_node']
    if i == 97:
        dev = DeviceTrie(cache, idx=op['idx'])
        print('n_nodes', dev.n_nodes, 'query', op['tokens'])
        for (mi, mo, mode, dl) in [(op['min_input_size'], op['min_output_size'], 'mix', 64), (0, 32, 'mix', 64), (1, 0, 'mix', 64), (0, 0, 'mix', 64),
                                   (0, 32, 'mix', 16), (0, 8, 'output', 64), (1, 0, 'input', 64), (0, 0, 'mix', 63)]:
            g = dev.hier_get([list(op['tokens'])], decoding_length=dl, branch_length=12, min_input_size=mi, min_output_size=mo, mode=mode)[0]
            h = cache.hier_get_packed(list(op['tokens']), decoding_length=dl, branch_length=12, min_input_size=mi, min_output_size=mo, mode=mode, idx=op['idx'])
            ok = g[0] == h[0].tolist() and [int(x) for x in g[1]] == [int(x) for x in h[1]] and g[2] == h[3]
            print((mi, mo, mode, dl), 'OK' if ok else 'BAD', 'dev n', len(g[0]), g[0][:6], g[2], '| host n', len(h[0]), h[0][:6].tolist(), h[3])
        break
