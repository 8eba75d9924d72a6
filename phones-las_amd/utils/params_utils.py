"""Hyper-parameter handling with the reference's semantics (utils/params_utils.py:12-172):
defaults, merge with ``model_dir/hparams.json`` (an existing file wins unless ``--reset``; the file is a
JSON *string* holding the JSON object, params_utils.py:28-30,89-90) and the encoder/decoder split."""
import json
import os

__all__ = ['HParams', 'create_hparams', 'get_default_hparams', 'get_encoder_decoder_hparams']


class HParams(object):
    """Minimal stand-in for tf.contrib.training.HParams with the del/pop helpers the reference adds."""

    def __init__(self, **kwargs):
        object.__setattr__(self, '_names', [])
        for k, v in kwargs.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name not in self._names:
            self._names.append(name)
        object.__setattr__(self, name, value)

    def set_hparam(self, name, value):
        if name not in self._names:
            raise KeyError(name)
        object.__setattr__(self, name, value)

    def del_hparam(self, name):
        if name in self._names:
            self._names.remove(name)
            object.__delattr__(self, name)

    def pop_hparam(self, name):
        value = getattr(self, name)
        self.del_hparam(name)
        return value

    def get_hparam(self, name):
        return getattr(self, name)

    def values(self):
        return {k: getattr(self, k) for k in self._names}

    def to_json(self):
        return json.dumps(self.values(), sort_keys=True)

    def save_to_file(self, filename):
        tmp = '%s.tmp.%d' % (filename, os.getpid())
        with open(tmp, 'w') as f:
            json.dump(self.to_json(), f)          # double encoding, as the reference does
        os.replace(tmp, filename)                 # readers never see a truncated file

    def __repr__(self):
        return 'HParams(%s)' % ', '.join('%s=%r' % (k, getattr(self, k)) for k in self._names)


def get_default_hparams():
    """utils/params_utils.py:33-77."""
    return HParams(
        learning_rate=1e-3, dropout=0.2, l2_reg_scale=1e-6, add_noise=0, noise_std=0.1, ctc_weight=-1.,
        tpu_name='', max_frames=-1, max_symbols=-1, num_channels=39,
        encoder_layers=3, encoder_units=64, use_pyramidal=True, unidirectional=False,
        decoder_layers=2, decoder_units=128, target_vocab_size=0, binf_count=0, embedding_size=0,
        sampling_probability=0.1, sos_id=1, eos_id=2, bottom_only=False, pass_hidden_state=False,
        decoding_length_factor=1.0, attention_type='luong', attention_layer_size=None, beam_width=0,
        binary_outputs=False, binf_sampling=False, binf_projection=False, binf_projection_reg_weight=1.0,
        binf_trainable=False, multitask=False, mapping=None)


def create_hparams(args, target_vocab_size=None, binf_count=None, sos_id=1, eos_id=2, write=True):
    """utils/params_utils.py:80-116.  write=False: read / merge only (the replicas of a multi-GPU job other than rank 0:
    one writer per model_dir)."""
    hparams = get_default_hparams()
    hparams_file = os.path.join(args.model_dir, 'hparams.json')
    is_reset = getattr(args, 'reset', False)
    if os.path.exists(hparams_file) and not is_reset:
        with open(hparams_file, 'r') as f:
            hparams_dict = json.loads(json.load(f))
        for name, value in vars(args).items():
            if name not in hparams_dict:
                hparams_dict[name] = value
    else:
        if target_vocab_size is None:
            raise ValueError('Target vocabulary size is not specified.')
        hparams_dict = dict(vars(args))
        hparams_dict.update({'sos_id': sos_id, 'eos_id': eos_id, 'target_vocab_size': target_vocab_size,
                             'binf_count': binf_count})
    for name in list(hparams.values()):
        value = hparams_dict.get(name, None)
        if value is not None:
            if name == 'mapping':
                if not isinstance(value, list):
                    with open(value, 'r') as f:
                        value = [int(x.strip()) for x in f]
                hparams.del_hparam(name)
                hparams.add_hparam(name, value)
            else:
                hparams.set_hparam(name, value)
    if write:
        os.makedirs(args.model_dir, exist_ok=True)
        hparams.save_to_file(hparams_file)
    return get_encoder_decoder_hparams(hparams)


def get_encoder_decoder_hparams(hparams):
    """utils/params_utils.py:119-172: split into params.encoder / params.decoder groups."""
    pop = hparams.pop_hparam
    learning_rate, ctc_weight, tpu_name = pop('learning_rate'), pop('ctc_weight'), pop('tpu_name')
    max_frames, max_symbols, dropout = pop('max_frames'), pop('max_symbols'), pop('dropout')
    l2_reg_scale, add_noise, noise_std, mapping = pop('l2_reg_scale'), pop('add_noise'), pop('noise_std'), pop('mapping')
    binary_outputs, binf_sampling, binf_projection = pop('binary_outputs'), pop('binf_sampling'), pop('binf_projection')
    binf_projection_reg_weight, binf_trainable = pop('binf_projection_reg_weight'), pop('binf_trainable')
    multitask, num_channels = pop('multitask'), pop('num_channels')
    encoder = HParams(num_layers=pop('encoder_layers'), num_units=pop('encoder_units'),
                      use_pyramidal=pop('use_pyramidal'), unidirectional=pop('unidirectional'), dropout=dropout)
    decoder = HParams(num_layers=pop('decoder_layers'), num_units=pop('decoder_units'), dropout=dropout,
                      binary_outputs=binary_outputs, binf_sampling=binf_sampling, binf_projection=binf_projection,
                      binf_projection_reg_weight=binf_projection_reg_weight, max_symbols=max_symbols,
                      multitask=multitask, binf_trainable=binf_trainable)
    for name, value in hparams.values().items():
        decoder.add_hparam(name, value)
    return HParams(learning_rate=learning_rate, mapping=mapping, l2_reg_scale=l2_reg_scale, add_noise=add_noise,
                   noise_std=noise_std, ctc_weight=ctc_weight, tpu_name=tpu_name, max_frames=max_frames,
                   max_symbols=max_symbols, num_channels=num_channels, encoder=encoder, decoder=decoder)
