"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/reference_model.py). Never imported by image2text_amd."""
