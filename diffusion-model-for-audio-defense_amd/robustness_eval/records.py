"""Per-example certification records in the reference's format (certified_robustness_eval.py:126-146):
a JSON list of {'id', 'y_true', 'y_pred', 'certified_radius'} at
`<save_path>/sigma=<sigma>/sigma=<sigma>_N=<num_sampling>.json`, rewritten after every batch (indent=4).
`resume=True` reloads an existing file so that an interrupted N = 100 000 run continues after its last record.
`append_batch(..., extra=[{...}, ...])` merges one dict of additional keys into each record (the driver's opt-in
`--audit` outcome); without it the records carry exactly the reference's four keys."""
import json
import os

__all__ = ['CertificationRecords']


class CertificationRecords:

    def __init__(self, save_path, sigma, num_sampling, resume=False):
        self.dir = os.path.join(save_path, 'sigma={}'.format(sigma))
        self.path = os.path.join(self.dir, 'sigma={}_N={}.json'.format(sigma, num_sampling))
        self.records = []
        if resume and os.path.exists(self.path):
            with open(self.path) as f:
                self.records = json.load(f)

    def __len__(self):
        return len(self.records)

    def append_batch(self, targets, y_certified, r_certified, extra=None):
        """Same fields and id numbering as the reference's loop body (id = running example index)."""
        total = len(self.records)
        for i in range(len(targets)):
            rec = {'id': i + total,
                   'y_true': int(targets[i]),
                   'y_pred': int(y_certified[i]),
                   'certified_radius': float(r_certified[i])}
            if extra is not None:
                rec.update(extra[i])
            self.records.append(rec)

    def flush(self):
        os.makedirs(self.dir, exist_ok=True)
        tmp = self.path + '.tmp'
        with open(tmp, 'w') as f:
            json.dump(self.records, f, indent=4)
        os.replace(tmp, self.path)          # a killed run never leaves a truncated file behind
