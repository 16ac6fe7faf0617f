const char afx_build_id_str[] = "394921446e70";
