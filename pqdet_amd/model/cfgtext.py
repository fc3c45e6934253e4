"""Darknet-style ``.cfg`` text reader (host side, pure Python).

Behavioural mirror of the reference lexer ``model/parser.py:256-359`` (``Parser.read/omit/match/
name/val/parse_line/gen_lines/gen_blocks/gen_layers``) and of ``str2value`` (``model/parser.py:114-120``)
and ``DEFAULT_LAYERS`` (``model/parser.py:43-104``).  The implementation here is a slice/regex based
scanner instead of a char-at-a-time state machine, but the observable results are the same:

* blank lines and lines whose first non-blank char is ``#`` are skipped; line numbers are 1-based;
* ``[section`` starts a layer (the closing ``]`` is never checked); unknown section -> ``KeyError``;
* ``key = v[,v...]  # comment``; every token goes through :func:`str2value`; *whitespace stays inside
  tokens* (``layers = -1, 46`` -> ``[-1, 46]`` only because ``int(' 46')`` accepts blanks); a single
  value stays scalar, anything with a comma becomes a list (a trailing comma yields a 1-element list);
* attributes before the first section are dropped; later sections' attributes overwrite defaults;
* errors surface as ``SyntaxError('line N: ...')`` with the reference's two message shapes.
"""
import re
from typing import IO, Dict, Iterator, List, Tuple, Union

# model/parser.py:43-104 -- per-section defaults (values are part of the cfg surface).
SECTION_DEFAULTS: Dict[str, dict] = {
    'net': dict(name='net', channels=3),
    'convolutional': dict(name='convolutional', filters=1, size=1, stride=1, pad=0, padding=0,
                          groups=1, activation='logistic', batch_normalize=0),
    'fc': dict(name='fc', input=1, output=1, activation='logistic'),
    'shortcut': dict(name='shortcut', activation='linear', alpha=1, beta=1),
    'scale_channels': dict(name='scale_channels'),
    'route': dict(name='route', layers=-1),
    'maxpool': dict(name='maxpool', size=1, stride=1, pad=0, padding=0),
    'avgpool': dict(name='avgpool', height=1, width=1),
    'upsample': dict(name='upsample', stride=2),
    'yolo': dict(name='yolo', classes=1, ignore_thresh=.5, bbox_loss='giou', l1_loss_gain=0.1),
    # the reference declares dropout without a 'name' default; it is unusable there too
    # (KeyError at model/parser.py:385) and we keep that behaviour.
    'dropout': dict(probability=0.5),
}

Scalar = Union[int, float, str]


def str2value(token: str) -> Scalar:
    """``model/parser.py:114-120``: int when there is no '.', else float, else the raw string."""
    try:
        return float(token) if '.' in token else int(token)
    except ValueError:
        return token


def _is_name_start(ch: str) -> bool:
    return ch.isalpha() or ch == '_'


def _scan_name(line: str, pos: int) -> Tuple[str, int]:
    """Longest identifier starting at ``pos`` (may be empty).  Returns (name, next_pos)."""
    end = pos
    n = len(line)
    while end < n and (_is_name_start(line[end]) or (end > pos and line[end].isdigit())):
        end += 1
    return line[pos:end], end


class _Missing(Exception):
    """Ran off the end of the line where the grammar needs another character."""


def _parse_value(line: str, pos: int) -> Union[Scalar, List[Scalar]]:
    hash_at = line.find('#', pos)
    body = line[pos:] if hash_at < 0 else line[pos:hash_at]
    if ',' not in body:
        return str2value(body)
    parts = body.split(',')
    if parts[-1] == '':      # trailing comma: the reference never emits the empty tail token
        parts.pop()
    return [str2value(p) for p in parts]


def _parse_line(line: str):
    """-> ('layer', name) | ('attr', key, value).  Raises _Missing / SyntaxError like the reference."""
    n = len(line)
    if line[0] == '[':
        if n == 1:
            raise _Missing
        name, _ = _scan_name(line, 1)
        return ('layer', name)
    key, pos = _scan_name(line, 0)
    if pos >= n:
        # identifier ran to end of line: the reference then tests the *last* char against '='
        pos = n - 1
    # skip blanks before '='
    while line[pos].isspace():
        pos += 1
        if pos >= n:
            raise _Missing
    if line[pos] != '=':
        raise SyntaxError("expect '{}', got '{}'".format('=', line[pos]))
    pos += 1
    if pos >= n:
        raise _Missing
    while line[pos].isspace():
        pos += 1
        if pos >= n:
            raise _Missing
    return ('attr', key, _parse_value(line, pos))


def iter_statements(fp: IO) -> Iterator[tuple]:
    for lineno, raw in enumerate(fp, 1):
        line = raw.strip()
        if not line or line[0] == '#':
            continue
        try:
            yield _parse_line(line)
        except _Missing:
            raise SyntaxError('line %d: something is missing' % lineno)
        except SyntaxError as e:
            raise SyntaxError(('line %d: ' % lineno) + str(e))


def iter_sections(fp: IO) -> Iterator[dict]:
    """Yield one dict per ``[section]`` = defaults overlaid with the file's attributes."""
    cur = None
    for st in iter_statements(fp):
        if st[0] == 'layer':
            if cur is not None:
                yield cur
            cur = SECTION_DEFAULTS[st[1]].copy()
        elif cur is not None:
            cur[st[1]] = st[2]
    if cur is not None:
        yield cur
