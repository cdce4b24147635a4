"""Certification driver on the MI355X engine: the flags, data flow and record format of the reference's
certified_robustness_eval.py (l.15-146), with the pieces of this package in place of the CUDA ones.

  python certified_robustness_eval.py --data_path <SC09 test folder> --victim_path <classifier .pth>
         --defender_path <DiffWave .pkl> --config configs/config.json --sigma 0.5 --num_sampling 100000
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 certified_robustness_eval.py ...

Differences from the reference, all additive:
  * one process per GPU under torch.distributed: every rank walks the same examples, RobustCertificate shards each
    example's N Monte Carlo samples over the ranks and all-reduces the int64[10] counts; rank 0 writes the records;
  * `--resume` continues after the last record of an existing JSON file (N = 100 000 takes about a minute per clip);
  * `--victim_path` is honoured (the reference ignores it and builds a sigma-dependent path, l.57-59); that path is
    still the default when the flag is not given.
`run(args)` is importable so that tests can drive it without a subprocess."""
import argparse
import os

import torch
from torch.utils.data import DataLoader

from audio_models.ConvNets_SpeechCommands.create_model import create_model
from datasets.sc_dataset import SC09Dataset
from diffusion_models.diffwave_ddpm import create_diffwave_model
from dmad_hip.transforms import MelSpectrogramDB
from robustness_eval.certified_robust import RobustCertificate
from robustness_eval.records import CertificationRecords
from transforms import FixAudioLength, LoadAudio


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # SC09 classifier arguments
    parser.add_argument("--data_path", default='datasets/speech_commands/test')
    parser.add_argument("--victim_path", default=None)
    parser.add_argument("--classifier_input", choices=['mel32'], default='mel32', help='input of NN')
    parser.add_argument("--num_per_class", type=int, default=10)
    # DiffWave arguments
    parser.add_argument('--config', type=str, default='configs/config.json', help='JSON file for configuration')
    parser.add_argument('--defender_path', type=str,
                        default='diffusion_models/DiffWave_Unconditional/exp/ch256_T200_betaT0.02/logs/checkpoint/1000000.pkl')
    # certified robust arguments
    parser.add_argument('--defense_method', type=str, default='diffusion', choices=['diffusion', 'randsmooth'])
    parser.add_argument('--sigma', type=float, default=0.25)
    parser.add_argument('--num_sampling', type=int, default=1000)
    # device arguments
    parser.add_argument("--dataload_workers_nums", type=int, default=8, help='number of workers for dataloader')
    parser.add_argument("--batch_size", type=int, default=16, help='batch size')
    parser.add_argument('--gpu', type=int, default=0)
    # file saving arguments
    parser.add_argument('--save_path', type=str, default='_Experiments/certified_robustness/records')
    # additions
    parser.add_argument('--resume', action='store_true', help='continue after the last record of an existing file')
    parser.add_argument('--noise_source', choices=['device', 'torch_cpu'], default='device')
    parser.add_argument('--calibrate_margins', type=int, default=1024,
                        help='exact-vote engine: samples per clip used to measure the recheck bounds for the loaded checkpoints on the first '
                             '--calibrate_clips clips of a sigma (bounds only widen, never below the engine defaults; 0 = engine defaults)')
    parser.add_argument('--calibrate_clips', type=int, default=3)
    parser.add_argument('--audit', type=int, default=0,
                        help='exact-vote engine: per example, re-evaluate this many samples that voted on the 16-bit tier on the split-f16 tier '
                             'and write the outcome into the record (key "audit")')
    return parser


def run(args, classifier=None, denoiser=None, log=print):
    """Returns the list of records.  `classifier` / `denoiser` may be passed ready-made (tests, synthetic weights)."""
    distributed = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1
    if distributed:
        local = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group('nccl')
        rank = torch.distributed.get_rank()
    else:
        torch.cuda.set_device(args.gpu)
        rank = 0

    if classifier is None:
        path = args.victim_path or ('audio_models/ConvNets_SpeechCommands/checkpoints/gaussian_aug_resnext29_8_64_sgd_plateau_'
                                    'bs50_lr1.0e-02_wd1.0e-02/sigma={}-best-acc.pth'.format(args.sigma))
        classifier = create_model(path)
    classifier.cuda()
    transform = _Compose([LoadAudio(), FixAudioLength()])
    test_dataset = SC09Dataset(folder=args.data_path, transform=transform, num_per_class=args.num_per_class)
    test_dataloader = DataLoader(test_dataset, batch_size=args.batch_size, sampler=None, shuffle=False,
                                 pin_memory=True, num_workers=args.dataload_workers_nums)
    if args.defense_method == 'diffusion' and denoiser is None:
        denoiser = create_diffwave_model(model_path=args.defender_path, config_path=args.config)
    if args.defense_method == 'randsmooth':
        denoiser = None
    RC = RobustCertificate(classifier=classifier, transform=MelSpectrogramDB(), denoiser=denoiser,
                           noise_source=args.noise_source, calibrate=getattr(args, 'calibrate_margins', 0),
                           calibrate_clips=getattr(args, 'calibrate_clips', 3), log=(log if rank == 0 else None))

    records = CertificationRecords(args.save_path, args.sigma, args.num_sampling, resume=args.resume)
    done, seen = len(records), 0
    for batch in test_dataloader:
        waveforms = torch.unsqueeze(batch['samples'], 1)
        targets = batch['target']
        n = waveforms.shape[0]
        if seen + n <= done:                     # already certified in an earlier run
            seen += n
            continue
        keep = slice(max(done - seen, 0), n)
        seen += n
        waveforms, targets = waveforms[keep].cuda(), targets[keep].cuda()
        n_audits = len(RC.audit_log)
        y_certified, r_certified = RC.certify(x=waveforms, y=targets, sigma=args.sigma, n_0=100, n=args.num_sampling,
                                              batch_size=args.batch_size, audit=getattr(args, 'audit', 0))
        audits = RC.audit_log[n_audits:]
        extra = None
        if getattr(args, 'audit', 0) > 0:       # an audit that was asked for leaves a trace in every record, also when it could not run
            extra = ([{'audit': a} for a in audits] if len(audits) == len(targets) else
                     [{'audit': None, 'audit_unavailable': 'needs the fused loop with device noise on an exact-vote engine'}] * len(targets))
        records.append_batch(targets.tolist(), y_certified.tolist(), r_certified.tolist(), extra=extra)
        if rank == 0:
            records.flush()
            msg = 'certified %d / %d examples' % (len(records), len(test_dataset))
            eng = getattr(denoiser, 'engine', None)
            if eng is not None and getattr(eng, 'precision', None) == 2:      # exact-vote engine: what the recheck tiers did so far (this rank)
                voted, left16, to_fp32 = eng.recheck_stats(detail=True)
                msg += '; of %d samples voted on this rank %d (%.2f %%) left the 16-bit tier and %d reached fp32 (bounds %.4g / %.3g)' % (
                    voted, left16, 100.0 * left16 / max(voted, 1), to_fp32, eng.recheck_margin, eng.recheck_margin2)
            log(msg)
    return records.records


if __name__ == '__main__':
    run(build_parser().parse_args())
