import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tonic_amd.collector import Block, Collector

def check(tag):
    try:
        torch.zeros(4, device='cuda'); torch.cuda.synchronize(); print('ok   ', tag)
    except Exception as e:
        print('ERROR', tag, str(e).splitlines()[0])

check('start')
block = Block(256, 28, 8); check('block')
first = Collector(block, 3); check('first create')
second = Collector(block, 3); check('second create')
second.close(); check('second close')
first.close(); check('first close')
again = Collector(block, 3); check('again create')
again.close(); check('again close')
many = Collector(Block(1280, 28, 8), 3); check('temp block create')
many.close(); check('temp close')
os.environ['TONIC_AMD_COLLECTOR_PUSH'] = '0'
off = Collector(Block(256, 28, 8), 3); check('off create')
off.close(); check('off close')
