"""Mirror of the reference's quant/ package: quant.Weight_Only and quant.Weight_Activation,
each with the modules Qop, Quantizer and Chunker and the same class names inside them."""
