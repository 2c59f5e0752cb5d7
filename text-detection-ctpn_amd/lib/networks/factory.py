"""get_network(name) with the reference's contract (lib/networks/factory.py:4-14): 'VGGnet_test' is the inference
net; 'VGGnet_train' is out of scope (training) and raises."""
from .VGGnet_test import VGGnet_test


def get_network(name):
    tag = name.split('_')
    if len(tag) >= 2 and tag[0] == 'VGGnet' and tag[1] == 'test':
        return VGGnet_test()
    if len(tag) >= 2 and tag[0] == 'VGGnet' and tag[1] == 'train':
        raise KeyError('VGGnet_train: training is out of scope of the MI355X inference path')
    raise KeyError('Unknown dataset: {}'.format(name))
