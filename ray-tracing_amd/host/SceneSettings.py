"""Host mirror of the reference's settings record (SceneSettings.py:1-5): the `set` line of a scene file.
Three attributes with the reference's names; the two counts are truncated to int the way its constructor does
(Parser.py hands them over as floats).  The renderer reads them through scene_pack / rt_scene_desc."""


def _count(value):
    # int() of a float truncates toward zero: `set ... 5.9 10.2` means 5 shadow rays per side, 10 levels
    return int(value)


class SceneSettings:
    __slots__ = ("background_color", "root_number_shadow_rays", "max_recursions")

    def __init__(self, background_color, root_number_shadow_rays, max_recursions):
        self.background_color = background_color
        self.root_number_shadow_rays, self.max_recursions = _count(root_number_shadow_rays), _count(max_recursions)

    def __repr__(self):
        return "SceneSettings(background=%s, shadow_rays=%d^2, max_recursions=%d)" % (
            list(self.background_color), self.root_number_shadow_rays, self.max_recursions)
