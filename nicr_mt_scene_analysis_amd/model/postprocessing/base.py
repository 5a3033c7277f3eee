"""Entry point the decoders call on their raw outputs (interface of reference
model/postprocessing/base.py:13-41): `postprocess(data, batch, is_training)` routes to the
training or the inference variant; a subclass that only defines the training variant uses it
for inference too."""


class PostprocessingBase:
    def postprocess(self, data, batch, is_training=True):
        """data: (output, side_outputs) of a decoder; batch: the collated input batch."""
        if is_training:
            return self._postprocess_training(data, batch)
        return self._postprocess_inference(data, batch)

    def _postprocess_training(self, data, batch):
        raise NotImplementedError(f'{type(self).__name__} must define _postprocess_training')

    def _postprocess_inference(self, data, batch):
        return self._postprocess_training(data, batch)
