"""Dataset base of the data path (drop-in for reference msmctts/datasets/base_dataset.py:24-286): same constructor
arguments, id-list / feature-path / book conventions, random segment selection and normalisation, so the ``dataset:``
section of a reference YAML works unchanged.

An utterance id is the tuple of whitespace-separated attributes of its id-list line; ``feature_path[i]`` is a template
formatted with those attributes (``'mels/{0}.npy'``), or an existing file holding all utterances (a "book":
``.list/.txt`` lines ``id|v v v|...`` with ``_``-joined vectors, ``.pkl``, ``.yaml``).  In training every item is a random
window of ``segment_length`` (in the unit of ``frameshift``) cut from each time-aligned feature; windows are read from
the files directly (``pre_load=False``) or from arrays loaded once at start-up.
"""
import math
import os
import pickle
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..utils.config import read_yaml
from . import readers

MIN_DATASET_SIZE = 3200          # a training "epoch" is at least this many items (small corpora are cycled)

_READERS = {'.npy': 'npy', '.dat': 'raw', '.mgc': 'raw', '.ap': 'raw', '.wav': 'wav', '.pt': 'pt'}


def feature_normalize(feature, stat, denormalize=False):
    """affine (and optionally min-max) normalisation described by a statistics YAML (reference utils/utils.py:190-203)"""
    if denormalize:
        feature = (feature - stat['shift']) / stat['scale']
    if stat['method'] == 'minmax':
        lo, hi = np.asarray(stat['min']), np.asarray(stat['max'])
        feature = (feature - lo) / (hi - lo) if not denormalize else (hi - lo) * feature + lo
    if not denormalize:
        feature = feature * stat['scale'] + stat['shift']
    return feature.astype(np.float32)


def align_features(feat_dict, frameshift):
    """trim the time-aligned features of one utterance to a common duration that every frame shift divides
    (reference utils/utils.py:161-187); durations further apart than 10 % raise"""
    seqs = {k: v for k, v in feat_dict.items() if k in frameshift and frameshift[k] > 0}
    if not seqs:
        return seqs
    durations = {k: 1.0 * v.shape[0] * frameshift[k] for k, v in seqs.items()}
    longest, shortest = max(durations.values()), min(durations.values())
    if longest / shortest >= 1.1:
        raise RuntimeError('files are unaligned seriously: %s' % durations)
    common = shortest - shortest % np.lcm.reduce([frameshift[k] for k in seqs])
    feat_dict.update({k: v[:int(shortest / frameshift[k])][:int(common / frameshift[k])] for k, v in seqs.items()})
    return feat_dict


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, id_list, feature, samplerate, dimension, frameshift, feature_path=None, feature_stat=None,
                 padding_value=None, segment_length=-1, pre_load=True, seed=1234, training=True):
        super().__init__()
        self.samplerate, self.feature = samplerate, feature
        self.dimension = {f: d for f, d in zip(feature, dimension) if d > 0}
        self.frameshift = {f: s for f, s in zip(feature, frameshift) if s is not None and s > 0}
        self.padding_value = ({f: v for f, v in zip(feature, padding_value)} if padding_value is not None
                              else {f: 0 for f in feature})
        self.segment_length, self.pre_load, self.training = segment_length, pre_load, training
        self.dataset = {}                                        # (utterance id, feature) -> array | path | string
        self.feature_stat = {}
        if feature_stat is not None:
            self.feature_stat = {f: read_yaml(p) for f, p in zip(feature, feature_stat) if p is not None}
        random.seed(seed)              # the python global generator, like the reference: shuffling and window starts
        self.id_list = self.prepare_dataset(id_list, feature_path)

    def __len__(self):
        return max(MIN_DATASET_SIZE, len(self.id_list)) if self.training else len(self.id_list)

    def __getitem__(self, index):
        return self.parse_case(index % len(self.id_list))

    # -- one item ------------------------------------------------------------------------------------------------
    def parse_case(self, index):
        uid = self.id_list[index]
        items = {f: self.dataset[(uid, f)] for f in self.feature if (uid, f) in self.dataset}
        dur, t0 = -1, 0.0
        if self.training and self.segment_length > 0:
            dur = self.segment_length
            ref = max(self.frameshift, key=self.frameshift.get)           # the coarsest feature fixes the window grid
            n = (items[ref].shape if self.pre_load else self.parse_file(items[ref], self.dimension[ref], return_shape=True))[0]
            last = max(0, n - math.ceil(dur / self.frameshift[ref]))
            t0 = 1.0 * random.randint(0, last) * self.frameshift[ref]
        for key, item in items.items():
            start, length = 0, -1
            if key in self.frameshift:
                start, length = int(t0 / self.frameshift[key]), int(dur / self.frameshift[key])
            if isinstance(item, (list, tuple, np.ndarray)):
                value = item[start:start + length if length > 0 else None]
            elif isinstance(item, str):
                read = self.parse_file if (os.path.isfile(item) or (':' in item and os.path.isfile(item.split(':', 1)[0]))) \
                    else self.parse_string
                value = read(item, dimension=self.dimension.get(key), start=start, length=length)
                if 0 in value.shape:
                    raise ValueError('Cannot parse string: {}'.format(item))
            else:
                raise TypeError('Unknown feature type: {}'.format(type(item)))
            if key in self.feature_stat:
                value = feature_normalize(value, self.feature_stat[key])
            items[key] = value
        if not self.training:
            items['_id'] = index
        return items

    # -- sources -------------------------------------------------------------------------------------------------
    def parse_file(self, path, dimension=None, start=0, length=-1, return_shape=False):
        kind = _READERS[os.path.splitext(path)[-1]]
        if kind == 'npy':
            return readers.read_npy(path, start, length, return_shape)
        if kind == 'raw':
            return readers.read_raw_float32(path, dimension, start, length, return_shape)
        if kind == 'wav':
            out = readers.read_wav(path, start, length, return_shape)
            return out if return_shape else out[0]
        return readers.read_torch(path, dimension, start, length, return_shape)

    def parse_string(self, string, dimension=None, start=0, length=-1, return_shape=False):
        x = np.array([float(v) for v in string.replace('_', ' ').split()])
        if dimension is not None:
            x = x.reshape(len(x) // dimension, dimension)
        if return_shape:
            return x.shape
        return x[start:length if length > 0 else None]            # (the reference's slice: end = length, not start + length)

    def parse_book(self, path, id_list=None, feat=None):
        ext = os.path.splitext(path)[-1]
        if ext in ('.list', '.txt'):
            book = {}
            with open(path) as f:
                for line in f:
                    cols = line.strip().split('|')
                    arrays = [np.array([float(v) if '_' not in v else [float(x) for x in v.split('_')] for v in col.split(' ')])
                              for col in cols[1:]]
                    book[cols[0]] = arrays if len(arrays) > 1 else arrays[0]
        elif ext == '.pkl':
            with open(path, 'rb') as f:
                book = pickle.load(f)
        elif ext == '.yaml':
            book = read_yaml(path)
        else:
            raise ValueError('unknown book format: %s' % path)
        for attrs in id_list or ():
            key = [a for a in attrs if a in book][0]
            self.dataset[(attrs, feat)] = np.asarray(book[key])
        return book

    def prepare_dataset(self, id_list_file, feature_path):
        if isinstance(id_list_file, (tuple, list)):                # several corpora: column i of every feature_path
            ids = []
            for i, one in enumerate(id_list_file):
                ids += self.prepare_dataset(one, [p[i] for p in feature_path])
            return ids
        if '.yaml' in id_list_file:                                # {utterance: {feature: item}}
            table = read_yaml(id_list_file)
            ids = sorted(table.keys())
            for uid in ids:
                for name, item in table[uid].items():
                    self.dataset[(uid, name)] = item
        else:
            with open(id_list_file) as f:
                ids = [tuple(line.strip().split()) for line in f.readlines()]
            for feat, path in zip(self.feature, feature_path):
                if isinstance(path, str) and os.path.isfile(path):
                    self.parse_book(path, id_list=ids, feat=feat)
                else:
                    self.dataset.update({(attrs, feat): path.format(*attrs) for attrs in ids})
        if self.pre_load and self.training:
            self.preload_files()
        if self.training:
            random.shuffle(ids)
        return ids

    def preload_files(self):
        for feat in self.feature:
            keys = [k for k in self.dataset if k[-1] == feat]
            if not keys or not isinstance(self.dataset[keys[0]], str):
                continue
            with ThreadPoolExecutor(max_workers=max(1, (os.cpu_count() or 2) // 2)) as pool:
                loaded = list(pool.map(lambda k: self.parse_file(self.dataset[k], self.dimension.get(feat)), keys))
            self.dataset.update(dict(zip(keys, loaded)))
