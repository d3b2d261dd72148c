"""Minimal stand-in for the reference's console output tree (superdsm/output.py:69-212, OUT OF SCOPE as a UI):
stages only need ``intermediate``, ``write`` and ``derive(muted=...)`` on whatever ``out`` they are handed."""
import sys


class Output:
    def __init__(self, muted=False, margin=0):
        self.muted = muted
        self.margin = margin

    def derive(self, muted=False, margin=0):
        return Output(self.muted or muted, self.margin + margin)

    def intermediate(self, line):
        pass            # transient status lines are not printed

    def write(self, line):
        if not self.muted:
            print(' ' * self.margin + str(line), file=sys.stdout)


class Text:
    BOLD = 'bold'

    @staticmethod
    def style(text, style):
        return text


def get_output(out=None):
    """``None`` -> console output, ``'muted'`` -> silent, an Output-like object -> itself."""
    if out is None:
        return Output()
    if isinstance(out, str):
        if out == 'muted':
            return Output(muted=True)
        raise ValueError(f'Unknown output: {out}')
    return out
