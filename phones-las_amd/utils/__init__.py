"""Host-side mirror of the reference's ``utils`` package (only what the hot path and its callers need)."""
from .params_utils import HParams, create_hparams  # noqa: F401
from .vocab_utils import UNK, SOS, EOS, UNK_ID, SOS_ID, EOS_ID, load_vocab, create_vocab_table  # noqa: F401
from .metrics_utils import edit_distance  # noqa: F401
from .features_utils import load_normalization  # noqa: F401
from .dataset_utils import input_fn, process_dataset, read_dataset  # noqa: F401
from .ipa_utils import load_binf2phone, get_mapping  # noqa: F401
