COMPILE = ["pir_api.cpp"]
DESCRIPTION = "the chunk loop's first remaining dimension from Coeff-form dim-0 results, as the reference does it: inverse transform, lift with the copy of the Q rows, forward transform of all rows (production: the Eval-form Q rows are kept)"
EDITS = [("pir_api.cpp", "                                      results, s, true));\n        return remaining_dimensions(ctx, dimensions, dimension_count, shape, chunks, results, remaining_query,\n                                    relinearization_key, out, s, true);",
          "                                      results, s, false));\n        return remaining_dimensions(ctx, dimensions, dimension_count, shape, chunks, results, remaining_query,\n                                    relinearization_key, out, s, false);")]
