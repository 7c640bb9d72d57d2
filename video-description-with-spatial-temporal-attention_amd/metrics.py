"""Caption sampling for evaluation: the decode half of `metrics.generate_sample_gpu_single_process`
(metrics.py:103-146).  Every video of a split is decoded with beam search, the hypothesis of minimal cost is kept
(:130-131), word ids become text (`_seqs2words`, :109-119) and the captions are written one per line to
`valid_samples.txt` / `test_samples.txt` -- the format of the reference's own `test/*.txt`.

Scoring those files (BLEU/METEOR/CIDEr through coco-caption, metrics.py:147-176) needs Java and the ground-truth
pickle and is out of scope; this module stops at the sample files and returns the captions."""
import os
from collections import OrderedDict

import numpy

MAXLEN = 50    # metrics.py:14


def seqs2words(caps, word_idict):
    """metrics.py:109-119: stop at the first 0 (<eos>); an id beyond the dictionary size prints as word_idict[1]."""
    out = []
    for cap in caps:
        words = []
        for w in cap:
            if w == 0:
                break
            words.append(word_idict[1] if w > len(word_idict) else word_idict[w])
        out.append(' '.join(words))
    return out


def build_sample_pairs(samples, vidIDs):
    """metrics.py:79-83: what coco-caption's scorer is fed -- {vidID: [{'image_id': vidID, 'caption': text}]}."""
    pairs = OrderedDict()
    for sample, vid in zip(samples, vidIDs):
        pairs[vid] = [{'image_id': vid, 'caption': sample}]
    return pairs


def sample_split(engine, model, f_init, f_next, options, whichset, beam=5, maxlen=MAXLEN, batched=False, tparams=None):
    """Best beam hypothesis per video of `whichset`, as word-id lists.  batched=False follows the reference loop
    (one gen_sample per video, metrics.py:123-133); batched=True decodes all videos together on the device
    (Attention.gen_sample_batch) and returns the same captions."""
    ctxgs, ctxg_masks, ctxls, ctxl_masks, ctxms, ctxm_masks = engine.prepare_data_for_blue(whichset)
    if batched:
        if not ctxgs:
            return []
        results = model.gen_sample_batch(tparams, options, numpy.asarray(ctxgs), numpy.asarray(ctxg_masks),
                                         numpy.asarray(ctxls), numpy.asarray(ctxms), k=beam, maxlen=maxlen)
        return [sample[int(numpy.argmin(score))] for sample, score in results]
    picked = []
    for ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask in zip(ctxgs, ctxg_masks, ctxls, ctxl_masks, ctxms, ctxm_masks):
        sample, score, _, _ = model.gen_sample(None, f_init, f_next, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask,
                                               options, None, beam, maxlen=maxlen)
        picked.append(sample[int(numpy.argmin(score))])
    return picked


def generate_sample_gpu_single_process(model_type, model_archive, options, engine, model, f_init, f_next,
                                       save_dir='./samples', beam=5, whichset='both', batched=False, tparams=None):
    """Same positional arguments as metrics.py:103-107.  Writes <save_dir>/valid_samples.txt and/or test_samples.txt and
    returns (samples_valid, samples_test) like the reference (:148-152): per split an OrderedDict vidID ->
    [{'image_id', 'caption'}] (build_sample_pairs); a split that was not requested, or that is empty, stays None / [] ."""
    os.makedirs(save_dir, exist_ok=True)
    samples = {'valid': None, 'test': None}
    for split in ('valid', 'test'):
        if whichset in (split, 'both'):
            ids = sample_split(engine, model, f_init, f_next, options, split, beam=beam, batched=batched, tparams=tparams)
            samples[split] = seqs2words(ids, engine.word_idict)
            with open(os.path.join(save_dir, '%s_samples.txt' % split), 'w') as f:
                f.write('\n'.join(samples[split]) + '\n')
    ids = {'valid': engine.valid_ids, 'test': engine.test_ids}
    return tuple(build_sample_pairs(samples[sp], ids[sp]) if samples[sp] else samples[sp] for sp in ('valid', 'test'))
