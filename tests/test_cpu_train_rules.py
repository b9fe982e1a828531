"""Host rules of train.py that no CLI run reaches in a few steps (reference train.py:124-125,178-197): NaN abort,
checkpoint-on-new-minimum, best-epoch selection with CONVERGENCE_TOLERANCE, early stopping."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'object-detection-yolov3_amd'))


def _reference_rule(test_loss, early_stopping_count):
    """The lines of the reference loop, restated: returns (save_checkpoint, best_epoch, stop)."""
    save = (len(test_loss) - 1) == np.argmin(test_loss)
    err = np.abs(np.asarray(test_loss) - np.min(test_loss))
    err[err < 1e-4] = 0
    best = np.where(err == 0)[0][0]
    return bool(save), int(best), bool(len(test_loss) - best > early_stopping_count)


def test_early_stopping_rule():
    import train
    series = [5.0, 4.0, 3.0, 3.00005, 2.99995, 3.2, 3.1, 2.9999, 3.5, 3.6, 3.7]
    for n in range(1, len(series) + 1):
        tl = series[:n]
        for patience in (1, 2, 3, 10):
            save, best, stop = _reference_rule(tl, patience)
            assert train.is_new_minimum(tl) == save
            assert train.best_epoch_of(tl) == best
            assert train.should_stop(tl, patience) == stop
    # within the tolerance the EARLIEST epoch wins, so tiny improvements do not reset the patience counter
    assert train.best_epoch_of([3.0, 2.99996, 2.99992]) == 0 and train.should_stop([3.0, 2.99996, 2.99992], 2)
    assert not train.should_stop([3.0], 1) and train.should_stop([3.0, 3.1], 1)
    assert train.is_new_minimum([3.0, 2.99995]) and not train.is_new_minimum([3.0, 3.0])      # argmin takes the first minimum


def test_nan_abort():
    import train
    train.abort_on_nan(1.5, 'x')
    train.abort_on_nan(np.float32(0.0), 'x')
    with pytest.raises(RuntimeError, match='Training Loss went to NaN'):
        train.abort_on_nan(float('nan'), 'Training Loss went to NaN, try a lower learning rate')
    with pytest.raises(RuntimeError, match='Test Loss went to NaN'):
        train.abort_on_nan(np.float32('nan'), 'Test Loss went to NaN')


def test_reader_count_is_capped_by_the_cores_one_rank_may_use():
    """8 ranks x 12 readers x 2 readers-per-rank on one host CPU must not happen by default or by flag."""
    from train import effective_reader_count, READER_COUNT
    assert effective_reader_count(None, 128, 1) == 12                 # one GPU, big host: the measured sweet spot
    assert effective_reader_count(None, 128, 8) == 12                 # 16 cores per rank
    assert effective_reader_count(None, 64, 8) == 6                   # 8 cores per rank: 7 usable, default share - 1
    assert effective_reader_count(None, 16, 8) == 1                   # 2 cores per rank
    assert effective_reader_count(None, 8, 1) == 6                    # this container
    assert effective_reader_count(3, 128, 8) == 3                     # the reference's value passes through
    assert effective_reader_count(48, 128, 8) == 15                   # an explicit request is capped too
    assert effective_reader_count(12, 8, 8) == 1 and effective_reader_count(0, 8, 1) == 1
    assert READER_COUNT == 3
