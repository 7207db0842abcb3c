"""Attribute-bag data transfer object (API of the reference's ``common/dto/Dto.py:1-44``).

Same public behaviour -- keyword construction, iteration over ``(name, value)`` pairs, a printable
fill-level tree and the ``_is_empty`` guard the models assert on -- written independently.  One
deliberate difference: ``_is_empty`` honours nested DTOs (the reference ignores the recursive
result); for the call sites on the hot path (all-``None`` leaves) both agree.
"""


class Dto(object):
    def __init__(self, **members):
        for key, value in members.items():
            setattr(self, key, value)

    def __iter__(self):
        return iter(list(vars(self).items()))

    def _is_empty(self):
        for _, value in self:
            if isinstance(value, Dto):
                if not value._is_empty():
                    return False
            elif value is not None:
                return False
        return True

    def __str__(self, indent=None):
        lines = []
        if indent is None:
            lines.append("Fill level of %s:" % object.__repr__(self))
            indent = ""
        for key in sorted(vars(self)):
            value = getattr(self, key)
            lines.append("%s[%s] %s" % (indent, " " if value is None else "x", key))
            if isinstance(value, Dto):
                lines.append(value.__str__(indent + "    ").rstrip("\n"))
        return "\n".join(lines) + "\n"
